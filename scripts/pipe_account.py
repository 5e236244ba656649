"""Developer tool (runs inside scripts/final_profile_gpu.sh): per-kernel pipe account from the two SQ passes + GRBM_GUI_ACTIVE of one build —
clock, matrix-pipe busy, waves parked (s_waitcnt / barrier), issue stalls, LDS / VALU / vector-memory instruction activity, MFMA + VALU co-execution.
Units (MI355X_MICROARCH.md, Per-instruction cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES and SQ_VALU_MFMA_COEXEC_CYCLES count cycles summed over the 1024 SIMDs; SQ_BUSY_CYCLES = 32 x kernel cycles (summed over the
chip's 32 shader engines); GRBM_GUI_ACTIVE = 8 x kernel cycles (summed over the XCDs).  usage: pipe_account.py <out dir> <commit>"""
import csv, glob, os, re, sys
from collections import defaultdict
out, commit = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "?"
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for sub in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")[:64]
            a = acc[n][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            if sub == "pmc_sq2" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                d = dur[n]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
print("commit", commit, "(profiled passes: the chip clocks lower than in the un-profiled bench line; shares are what to read)")
print(f"{'kernel':64s} {'us':>8s} {'GHz':>5s} {'MFMA busy':>9s} {'co-exec':>8s} {'parked':>7s} {'issue st.':>9s} {'LDS st.':>8s} {'VALU act':>8s} {'LDS act':>8s} {'FLAT act':>8s}")
for n, cs in sorted(acc.items(), key=lambda kv: -dur[kv[0]][0]):
    if not any(t in n for t in ("attn_", "gemm256")) or not dur[n][1]:
        continue
    g = lambda c: (cs[c][0] / cs[c][1]) if c in cs and cs[c][1] else float("nan")
    us = dur[n][0] / dur[n][1] / 1e3
    kc = g("SQ_BUSY_CYCLES") / 32.0
    wc = g("SQ_WAVE_CYCLES")
    pct = lambda x: f"{100.0 * x:6.1f} %"
    print(f"{n:64s} {us:8.1f} {g('GRBM_GUI_ACTIVE') / 8.0 / (us * 1e3):5.2f} {pct(g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / kc):>9s} {pct(g('SQ_VALU_MFMA_COEXEC_CYCLES') / 1024.0 / kc):>8s} "
          f"{pct(g('SQ_WAIT_ANY') / wc):>7s} {pct(g('SQ_WAIT_INST_ANY') / wc):>9s} {pct(g('SQ_WAIT_INST_LDS') / wc):>8s} {pct(g('SQ_ACTIVE_INST_VALU') / wc):>8s} {pct(g('SQ_ACTIVE_INST_LDS') / wc):>8s} {pct(g('SQ_ACTIVE_INST_FLAT') / wc):>8s}")
print("columns: us = average launch duration in the counter pass; GHz = GRBM_GUI_ACTIVE / 8 / duration; MFMA busy / co-exec = share of a SIMD's cycles with the matrix pipe busy /")
print("with matrix and vector instructions executing together; parked (SQ_WAIT_ANY), issue stalls (SQ_WAIT_INST_ANY), LDS stalls, VALU / LDS / FLAT (global loads, LDS-DMA) instruction activity = shares of the waves' cycles")
