"""Developer tool (GPU, make DEV=1): timing-only ablations of the MX GEMM main loop (prio_mode 4 / 5 / 6: no fragment reads / no DMA / no  [needs a developer build: make -C gliclass/c_amd DEV=1 (timing-only builds live in csrc/dev/gemm256x_dev.hip)]
MFMAs; 7: the traffic and MFMA format fp6 cross terms would have — 7/8 of the DMA, 24-byte scaled fragments, e2m3 MFMAs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
names = {1: "full", 8: "f16 part as 16x16x32"}
for rnd in range(2):
    for (name, M_, N, K, ep) in (("ffn2-as-bias", 65536, 768, 3072, 0), ("ffn1-as-bias", 65536, 3072, 768, 0), ("qkv-as-bias", 65536, 2304, 768, 0), ("c5-half-gate-up-as-bias", 32768, 8960, 1536, 0)):
        r = {pm: e.L.glc_debug_gemm_bench(e.h, M_, N, K, ep, 10, 100 * (1 + pm) + 9) for pm in names}
        print(f"r{rnd} {name:24s} " + "  ".join(f"{names[pm]} {r[pm]*1e3:7.1f} us" for pm in r), flush=True)
for pm in (7,):
    e.L.glc_debug_gemm_bench(e.h, 65536, 768, 3072, 0, 5, 100 * (1 + pm) + 10)
e.close()
