"""Developer tool (GPU, make DEV=1): the W128 loop (one wave per SIMD, 128 x 128 per wave) against the product loop, bias epilogue, short and long K."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
M = 65536
for rnd in range(2):
    for (name, N, K, ep) in (("n768-k768-bias", 768, 768, 0), ("n768-k3072-bias", 768, 3072, 0), ("n768-k12288-bias", 768, 12288, 0), ("n3072-k768-bias", 3072, 768, 0)):
        r = {which: e.L.glc_debug_gemm_bench(e.h, M, N, K, ep, 10, which) for which in (9, 13, 14)}
        print(f"r{rnd} {name:20s} 32x32 {r[9]*1e3:7.1f} us   Z16 {r[13]*1e3:7.1f} us ({r[9]/r[13]:.3f}x)   W128 {r[14]*1e3:7.1f} us ({r[9]/r[14]:.3f}x)", flush=True)
e.close()
