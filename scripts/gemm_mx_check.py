"""Developer tool (GPU): the MX cross-term GEMM (gemm256x.hip) — numerics against the split-f16 GEMM, then timing and the stamped phase account."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
out = (C.c_double * 5)()
MODES = {0: "bias, plain out", 1: "gelu + LN fold, row out", 2: "resid (raw rows, LN on the fly), raw rows + partials out", 3: "resid, plain out", 4: "qkv + LN fold, units"}
for mode in range(5):
    for (M, N, K, aa, wa) in ((256, 256, 32, 1.0, 0.05), (512, 768, 768, 2.0, 0.05), (1024, 768, 3072, 1.0, 0.1)) if mode else ((256, 256, 32, 1.0, 0.05), (256, 256, 64, 1.0, 0.05), (512, 768, 768, 2.0, 0.05), (1024, 3072, 768, 3.0, 0.02), (1024, 768, 3072, 1.0, 0.1), (512, 512, 1024, 100.0, 1.0)):
        if mode == 4: N = 768
        rc = e.L.glc_debug_gemm_mx_check(e.h, M, N, K, aa, wa, mode, out)
        print(f"check mode {mode} ({MODES[mode]}) M={M} N={N} K={K} a~U(+-{aa}) w~U(+-{wa}): rc={rc} max|mx-gs| {out[0]:.3e} (max |gs| {out[1]:.3e}) rel rms {out[2]/max(out[3],1e-30):.2e}" + (f" ln_part diff {out[4]:.2e}" if mode == 2 else "") + ("" if rc == 0 else "  " + e.L.glc_last_error().decode()), flush=True)
if os.environ.get("GLC_CHECK_ONLY"): e.close(); sys.exit(0)
EPI = {"bias": 0, "gelu": 1, "resid": 2}
M = 65536
shapes = [("attn-out", M, 768, 768, "resid"), ("ffn1", M, 3072, 768, "gelu"), ("ffn2", M, 768, 3072, "resid"), ("qkv-as-bias", M, 2304, 768, "bias")]
for rnd in range(2):
    for (name, M_, N, K, ep) in shapes:
        r = {}
        for which in (6, 9):
            r[which] = e.L.glc_debug_gemm_bench(e.h, M_, N, K, EPI[ep], 10, which)
        print(f"r{rnd} {name:12s} split-f16 {r[6]*1e3:7.1f} us   MX {r[9]*1e3:7.1f} us  ({r[6]/r[9]:.3f}x)  {2.0*M_*N*K/r[9]/1e9:7.1f} TF fp32-equivalent", flush=True)
for (M_, N, K) in ((65536, 3072, 768), (65536, 768, 3072)):
    e.L.glc_debug_gemm_bench(e.h, M_, N, K, 0, 5, 10)
e.close()
