"""Developer tool (GPU): the MX cross-term GEMM on GY rows (e2m3 parts with per-block scales, glc_common.h) — numerics of every epilogue  [needs a developer build: make -C gliclass/c_amd DEV=1 (GY images live in csrc/dev/gemm256x_dev.hip)]
against the split-f16 GEMM, then timing against the same kernel on GX rows (e4m3 parts)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
out = (C.c_double * 5)()
MODES = {0: "bias, plain out", 1: "gelu + LN fold, row out", 2: "resid (raw rows, LN on the fly), raw rows + partials out", 3: "resid, plain out", 4: "qkv + LN fold, units"}
bad = 0
for mode in range(5):
    for (M, N, K, aa, wa) in ((256, 256, 64, 1.0, 0.05), (512, 768, 768, 2.0, 0.05), (1024, 768, 3072, 1.0, 0.1), (512, 512, 1024, 100.0, 1.0)):
        if mode == 4: N = 768
        for fmt in (0, 10):
            rc = e.L.glc_debug_gemm_mx_check(e.h, M, N, K, aa, wa, mode + fmt, out)
            rel = out[2] / max(out[3], 1e-30)
            print(f"check mode {mode} ({MODES[mode]}) {'GY' if fmt else 'GX'} M={M} N={N} K={K} a~U(+-{aa}) w~U(+-{wa}): rc={rc} max|mx-gs| {out[0]:.3e} (max |gs| {out[1]:.3e}) rel rms {rel:.2e}" + (f" ln_part diff {out[4]:.2e}" if mode == 2 else "") + ("" if rc == 0 else "  " + e.L.glc_last_error().decode()), flush=True)
            if rc != 0 or not (rel < 5e-5): bad += 1
print("numerics:", "OK" if bad == 0 else f"{bad} FAILED")
if os.environ.get("GLC_CHECK_ONLY"): e.close(); sys.exit(1 if bad else 0)
EPI = {"bias": 0, "gelu": 1, "resid": 2}
M = 65536
shapes = [("attn-out", M, 768, 768, "resid"), ("ffn1", M, 3072, 768, "gelu"), ("ffn2", M, 768, 3072, "resid"), ("qkv-as-bias", M, 2304, 768, "bias"), ("c5-half-gate-up-as-bias", 32768, 8960, 1536, "bias")]
for rnd in range(2):
    for (name, M_, N, K, ep) in shapes:
        r = {which: e.L.glc_debug_gemm_bench(e.h, M_, N, K, EPI[ep], 10, which) for which in (9, 11)}
        print(f"r{rnd} {name:24s} GX {r[9]*1e3:7.1f} us   GY {r[11]*1e3:7.1f} us  ({r[9]/r[11]:.3f}x)  {2.0*M_*N*K/r[11]/1e9:7.1f} TF fp32-equivalent", flush=True)
for which in (10, 12):
    e.L.glc_debug_gemm_bench(e.h, 65536, 768, 3072, 0, 5, which)
e.close()
sys.exit(1 if bad else 0)
