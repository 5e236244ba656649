"""Developer tool (GPU): phase-by-phase cycle account of the 256-tile GEMM main loop (stamped build, glc_debug_gemm_bench which = 7 / 8)."""  [needs a developer build: make -C gliclass/c_amd DEV=1 (stamped builds)]
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
for which in (8, 7):
    for (M, N, K) in ((65536, 3072, 768), (65536, 768, 3072), (65536, 768, 768), (4096, 4096, 4096)):
        for _ in range(2):
            ms = e.L.glc_debug_gemm_bench(e.h, M, N, K, 0, 10, which)
        print(f"which={which} M={M} N={N} K={K}: {ms*1e3:.1f} us", flush=True)
e.close()
