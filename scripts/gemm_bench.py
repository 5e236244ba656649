"""Developer tool (GPU): GEMM microbenchmark matrix through glc_debug_gemm_bench."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["tiny"]
e = Engine(cfg, weights.make_weights(cfg, 1), dtype=sys.argv[1] if len(sys.argv) > 1 else "f16")
EPI = {"bias": 0, "gelu": 1, "resid": 2}
shapes = [(65536, 768, 768, "resid"), (65536, 768, 3072, "resid"), (65536, 3072, 768, "gelu"), (65536, 3072, 768, "bias"), (65536, 2304, 768, "bias"),
          (4096, 768, 768, "bias"), (4096, 3072, 768, "bias"), (4096, 768, 3072, "bias"), (8192, 8192, 8192, "bias"), (4096, 4096, 4096, "bias"),
          (65536, 768, 3072, "bias"), (16384, 3072, 3072, "bias")]
for (M, N, K, ep) in shapes:
    for which in (0, 1):
        ms = e.L.glc_debug_gemm_bench(e.h, M, N, K, EPI[ep], 10, which)
        if ms < 0:
            print("ERR", e.L.glc_last_error().decode()); continue
        print(f"M={M:6d} N={N:5d} K={K:5d} {ep:6s} {'tile256' if which == 0 else 'tile128'}  {ms:8.4f} ms  {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
e.close()
