"""Summarise rocprofv3 CSVs written by scripts/profile_gpu.sh: per-kernel time (kernel-trace stats)
and per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes.  gfx950 corrections per
MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2;
WRITE_SIZE is exact; both counters are in KiB."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:110]


def find(pattern):
    fs = glob.glob(os.path.join(out, pattern), recursive=True)
    return fs[0] if fs else None


st = find("trace/**/*kernel_stats.csv")
if st:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(st)))
    print("columns:", list(rows[0].keys()) if rows else None)
    for r in rows[:16]:
        name = r.get("Name") or r.get("KernelName") or ""
        print(f'{short(name):92s} calls={r.get("Calls")} avg_us={float(r.get("AverageNs", 0))/1e3:9.1f} total_ms={float(r.get("TotalDurationNs", 0))/1e6:9.2f} pct={r.get("Percentage")}')

traffic = {}
for tag, corr in (("fetch", 2.0), ("write", 1.0)):
    f = find(f"pmc_{tag}/**/*counter_collection.csv")
    if not f:
        print(f"no counter file for {tag}")
        continue
    acc = defaultdict(lambda: [0.0, 0])
    cols = None
    for r in csv.DictReader(open(f)):
        cols = cols or list(r.keys())
        name = r.get("Kernel_Name") or r.get("KernelName") or ""
        val = float(r.get("Counter_Value") or 0)
        a = acc[short(name)]
        a[0] += val
        a[1] += 1
    print(f"== {tag.upper()}_SIZE per launch (KiB raw -> bytes x1024, gfx950 correction x{corr}) ==")
    print("columns:", cols)
    for name, (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"{name:92s} launches={n:5d} avg_MB={tot / n * 1024 * corr / 1e6:10.2f}")
    for name, (tot, n) in acc.items():
        traffic.setdefault(name, {})[tag + "_bytes_per_launch"] = tot / n * 1024 * corr

# bench.py reads this (committed, merged per dtype, as profiles/traffic.json) for roofline.traffic of its dominant kernel class
import json
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
config = sys.argv[3] if len(sys.argv) > 3 else "base:64:1024"
if dtype == "f32":      # default mode: split-f16 attention (workgroup-shared kernel) and group-split GEMMs (gemm256s GS = last template flag 1)
    # (round 3: the MX cross-term GEMM gemm256x_kernel<EPI, VMODE, ...> where the MX pipeline runs, else gemm256s GS)
    cls = {"attention": r"attn_mx2?_kernel|attn_wg_kernel(IfLb1|<float, true)", "gemm_ffn1_gelu": r"gemm256x_kernel(ILi1ELb0|<1, false)|gemm256s_kernelIDF16_Li1ELb0ELb1|gemm256s_kernel<_Float16, 1, false, true",
           "gemm_qkv": r"gemm256x_kernel(ILi3ELb0|<3, false)|gemm256s_kernelIDF16_Li3ELb0ELb1|gemm256s_kernel<_Float16, 3, false, true",
           "gemm_ffn2": r"gemm256x_kernel(ILi2ELb0|<2, false)|gemm256s_kernelIDF16_Li2ELb0ELb1|gemm256s_kernel<_Float16, 2, false, true"}
else:
    cls = {"attention": r"attn_band_kernel", "gemm_ffn1_gelu": r"gemm256s?_kernelIDF16b?_Li1ELb0ELb0", "gemm_qkv": r"gemm256s?_kernelIDF16b?_Li3ELb0ELb0",
           "gemm_ffn2": r"gemm256s?_kernelIDF16b?_Li2ELb0ELb0"}
outj = {}
for c, pat in cls.items():
    for name, v in traffic.items():
        if re.search(pat, name) and "fetch_bytes_per_launch" in v and "write_bytes_per_launch" in v:
            outj[c] = dict(kernel=name, hbm_bytes_per_launch=v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"], **v)
commit = sys.argv[4] if len(sys.argv) > 4 else None
print("commit", commit)
json.dump({dtype: dict(config=config, source=os.path.basename(out.rstrip("/")), commit=commit,
                       note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB->bytes, FETCH x2 (gfx950)", kernels=outj)},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)
