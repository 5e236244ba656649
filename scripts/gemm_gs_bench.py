"""Developer tool (GPU): the group-split fp32-mode GEMM (gemm256s.hip, GS) on the c3 layer shapes through glc_debug_gemm_bench(which=6)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["tiny"]
e = Engine(cfg, weights.make_weights(cfg, 1), dtype="f16")
EPI = {"bias": 0, "gelu": 1, "resid": 2}
M = int(os.environ.get("GLC_M", 65536))
shapes = [("attn-out", M, 768, 768, "resid"), ("ffn1", M, 3072, 768, "gelu"), ("ffn2", M, 768, 3072, "resid"), ("qkv-as-bias", M, 2304, 768, "bias")]
for rep in range(2):
    for (name, M_, N, K, ep) in shapes:
        ms = e.L.glc_debug_gemm_bench(e.h, M_, N, K, EPI[ep], 10, 6)
        if ms < 0:
            print("ERR", e.L.glc_last_error().decode()); continue
        print(f"{name:12s} M={M_:6d} N={N:5d} K={K:5d} {ep:6s} {ms*1e3:8.1f} us  {2.0*M_*N*K/ms/1e9:7.1f} TF fp32-equivalent ({6.0*M_*N*K/ms/1e9:7.1f} TF MFMA rate)", flush=True)
e.close()
