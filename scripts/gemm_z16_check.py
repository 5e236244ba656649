"""Developer tool (GPU): the MX cross-term GEMM with its main loop on the 16 x 16 MFMA shapes (v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4,  [needs a developer build: make -C gliclass/c_amd DEV=1 (the 16 x 16 shapes live in csrc/dev/gemm256x_dev.hip)]
GemmArgs::z16) — numerics of every epilogue against the split-f16 GEMM, then timing against the 32 x 32 loop on the same GX images."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
out = (C.c_double * 5)()
MODES = {0: "bias, plain out", 1: "gelu + LN fold, row out", 2: "resid (raw rows, LN on the fly), raw rows + partials out", 3: "resid, plain out", 4: "qkv + LN fold, units"}
FMT = int(os.environ.get('GLC_FMT', 20)); WHICH = 14 if FMT == 30 else 13
bad = 0
for mode in range(5):
    for (M, N, K, aa, wa) in ((256, 256, 64, 1.0, 0.05), (512, 768, 768, 2.0, 0.05), (1024, 768, 3072, 1.0, 0.1), (512, 512, 1024, 20.0, 1.0)):
        if mode == 4: N = 768
        res = {}
        for fmt in (0, FMT):
            rc = e.L.glc_debug_gemm_mx_check(e.h, M, N, K, aa, wa, mode + fmt, out)
            res[fmt] = (rc, out[0], out[2] / max(out[3], 1e-30), out[4])
        ok = res[FMT][0] == 0 and res[FMT][2] < 3e-5
        bad += 0 if ok else 1
        print(f"check mode {mode} ({MODES[mode]}) M={M} N={N} K={K}: 32x32 rel rms {res[0][2]:.2e}   16x16 rc={res[FMT][0]} max|mx-gs| {res[FMT][1]:.3e} rel rms {res[FMT][2]:.2e}" + (f" ln_part diff {res[FMT][3]:.2e}" if mode == 2 else "") + ("" if ok else "  <-- " + e.L.glc_last_error().decode()), flush=True)
print("numerics:", "OK" if bad == 0 else f"{bad} FAILED")
if os.environ.get("GLC_CHECK_ONLY"): e.close(); sys.exit(1 if bad else 0)
EPI = {"bias": 0, "gelu": 1, "resid": 2}
M = 65536
shapes = [("attn-out", M, 768, 768, "resid"), ("ffn1", M, 3072, 768, "gelu"), ("ffn2", M, 768, 3072, "resid"), ("qkv-as-bias", M, 2304, 768, "bias"), ("c5-half-gate-up-as-bias", 32768, 8960, 1536, "bias")]
for rnd in range(2):
    for (name, M_, N, K, ep) in shapes:
        r = {which: e.L.glc_debug_gemm_bench(e.h, M_, N, K, EPI[ep], 10, which) for which in (9, WHICH)}
        print(f"r{rnd} {name:24s} 32x32 {r[9]*1e3:7.1f} us   16x16 {r[WHICH]*1e3:7.1f} us  ({r[9]/r[WHICH]:.3f}x)  {2.0*M_*N*K/r[WHICH]/1e9:7.1f} TF fp32-equivalent", flush=True)
e.close()
sys.exit(1 if bad else 0)
