"""Developer tool (GPU): wave-priority policies of the MX GEMM main loop (gemm256x.hip), interleaved same-process rounds + stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
EPI = {"bias": 0, "gelu": 1, "resid": 2}
M = 65536
shapes = [("attn-out", M, 768, 768, "resid"), ("ffn1", M, 3072, 768, "gelu"), ("ffn2", M, 768, 3072, "resid"), ("qkv-as-bias", M, 2304, 768, "bias")]
for rnd in range(3):
    for (name, M_, N, K, ep) in shapes:
        r = {}
        for pm in (0, 1, 2, 3):
            r[pm] = e.L.glc_debug_gemm_bench(e.h, M_, N, K, EPI[ep], 10, 100 * (1 + pm) + 9)
        print(f"r{rnd} {name:12s} " + "  ".join(f"prio{pm} {r[pm]*1e3:7.1f} us" for pm in r), flush=True)
for pm in (0, 2, 3):
    e.L.glc_debug_gemm_bench(e.h, 65536, 3072, 768, 0, 5, 100 * (1 + pm) + 10)
e.close()
