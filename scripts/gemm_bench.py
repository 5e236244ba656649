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
shapes = [(65536, 768, k, ep) for k in (256, 768, 1536, 3072, 6144) for ep in ("bias", "resid")] + \
         [(65536, 3072, k, ep) for k in (256, 768, 1536, 3072) for ep in ("bias", "gelu")] + [(4096, 4096, 4096, "bias"), (8192, 8192, 8192, "bias")]
for (M, N, K, ep) in shapes:
    ms = e.L.glc_debug_gemm_bench(e.h, M, N, K, EPI[ep], 10, int(os.environ.get('GLC_WHICH', '4' if os.environ.get('GLC_STAMPS') else '3')))
    if ms < 0:
        print("ERR", e.L.glc_last_error().decode()); continue
    tiles_per_cu = (M // 256) * (N // 256) / 256.0
    print(f"M={M:6d} N={N:5d} K={K:5d} {ep:6s} {ms:8.4f} ms {2.0*M*N*K/ms/1e9:8.1f} TF   per-tile {ms*1e3/tiles_per_cu:7.2f} us  per-Ktile {ms*1e3/tiles_per_cu/(K/64):6.3f} us", flush=True)
e.close()
