"""Developer tool (GPU): full-line vs half-line ring stages of the 256-tile GEMM (gemm256s.hip, round 3) — same process, interleaved.
1. whole forwards (base, B x S) in the default mode and the f16 mode, logits compared BITWISE between the two loops;
2. glc_debug_gemm_bench on the c3 layer shapes, group-split (which = 6) and f16 (which = 5), three interleaved rounds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth, weights, _lib
from gliclass.c_amd.engine import Engine
L = _lib.hip()
B, S = int(os.environ.get("GLC_B", 64)), int(os.environ.get("GLC_S", 1024))
cfg = CONFIGS["base"]
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=1234)
for dt in ("f32", "f16"):
    e = Engine.from_spec(cfg, "synthetic:base:42", dtype=dt)
    e.set_length_buckets(1)
    out = {}
    for fl in (1, 0, 1):
        L.glc_debug_set_gemm_full_lines(fl)
        got = e.forward(ids, mask)
        out.setdefault(fl, got.copy())
        e.profile(True)
        for _ in range(3): e.forward(ids, mask)
        pr = e.profile_read(); e.profile(False)
        per = {k: v[0] / max(v[1], 1) for k, v in pr.items() if v[1]}
        print(f"[{dt}] full_lines={fl}: " + "  ".join(f"{k} {per[k]*1e3:.0f}us" for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn1_gelu", "gemm_ffn2", "attention") if k in per), flush=True)
    same = np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))
    print(f"[{dt}] logits bitwise identical between the two loops: {same}; max |diff| {np.abs(out[0]-out[1]).max():.3e}", flush=True)
    e.close()
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
EPI = {"bias": 0, "gelu": 1, "resid": 2}
M = 65536
shapes = [("attn-out", M, 768, 768, "resid"), ("ffn1", M, 3072, 768, "gelu"), ("ffn2", M, 768, 3072, "resid"), ("qkv-as-bias", M, 2304, 768, "bias"), ("square", 4096, 4096, 4096, "bias")]
for which, label, mul in ((6, "group-split", 3.0), (5, "f16", 1.0)):
    for rnd in range(3):
        for (name, M_, N, K, ep) in shapes:
            r = {}
            for fl in (0, 1):
                L.glc_debug_set_gemm_full_lines(fl)
                r[fl] = e.L.glc_debug_gemm_bench(e.h, M_, N, K, EPI[ep], 10, which)
            print(f"{label:11s} r{rnd} {name:12s} half-line {r[0]*1e3:7.1f} us  full-line {r[1]*1e3:7.1f} us  ({r[0]/r[1]:.3f}x)  {2.0*mul*M_*N*K/r[1]/1e9:7.1f} TF MFMA rate", flush=True)
L.glc_debug_set_gemm_full_lines(1)
e.close()
