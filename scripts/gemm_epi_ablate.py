"""Developer tool (GPU): timing-only ablations of the MX GEMM's GX-row epilogue (GemmArgs::epi_abl through glc_debug_gemm_bench):
no stores at all / every tile stores into the same 256 rows (cache-resident target) / temporal instead of non-temporal stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
names = {0: "full", 1: "no stores", 2: "stores into 256 rows", 3: "temporal stores"}
for rnd in range(2):
    for (name, M_, N, K, ep) in (("ffn1-as-bias", 65536, 3072, 768, 0), ("qkv-as-bias", 65536, 2304, 768, 0), ("attn-out-as-bias", 65536, 768, 768, 0), ("ffn2-as-bias", 65536, 768, 3072, 0)):
        r = {a: e.L.glc_debug_gemm_bench(e.h, M_, N, K, ep, 10, 1000 * a + 9) for a in names}
        print(f"r{rnd} {name:18s} " + "  ".join(f"{names[a]} {r[a]*1e3:7.1f} us" for a in r), flush=True)
e.close()
