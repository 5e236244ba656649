"""Developer tool (GPU): band-attention microbenchmark on the c3 shape through glc_debug_attn_bench.
usage: attn_bench.py [variants, comma separated] [stamps 0/1]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth, weights
from gliclass.c_amd.engine import Engine
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
stamps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B, S = int(os.environ.get("GLC_B", 64)), int(os.environ.get("GLC_S", 1024))
cfg = CONFIGS["base"]
e = Engine(cfg, weights.make_weights(cfg, 42), dtype="f16")
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3, ragged=bool(os.environ.get("GLC_RAGGED")))
e.forward(ids, mask)
flops = 3 * 2.0 * B * cfg.heads * S * S * 64
for rep in range(2):
    for v in variants:
        cs = (ctypes.c_double * 2)()
        ms = e.L.glc_debug_attn_bench(e.h, 20, v, stamps if rep == 0 else 0, cs)
        if ms < 0:
            print("ERR", e.L.glc_last_error().decode()); continue
        print(f"variant {v}: {ms:.4f} ms  {flops/ms/1e9:7.1f} TF   checksum {cs[0]:.6e} {cs[1]:.6e}", flush=True)
e.close()
