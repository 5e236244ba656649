"""Developer tool (GPU): band-attention microbenchmark on the c3 shape through glc_debug_attn_bench, interleaved rounds in ONE process.
usage: attn_bench.py [variants, comma separated] [stamps 0/1] [dtype f16|bf16|f32]
variant bits: 2 = one wave per SIMD (LDS padding, per-wave kernel), 4 = the workgroup-shared kernel (attention_wg.hip)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,4").split(",")]
stamps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
B, S = int(os.environ.get("GLC_B", 64)), int(os.environ.get("GLC_S", 1024))
cfg = CONFIGS[os.environ.get("GLC_CONFIG", "base")]
e = Engine.from_spec(cfg, f"synthetic:{cfg.name}:42", dtype=dtype)
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3, ragged=bool(os.environ.get("GLC_RAGGED")))
e.set_length_buckets(1)
e.forward(ids, mask)
P = 2 * cfg.att_span
flops = B * S * (4.0 * S * cfg.hidden + 4.0 * P * cfg.hidden)          # QK^T + PV + c2p + p2c
for rep in range(3):
    for v in variants:
        cs = (ctypes.c_double * 2)()
        ms = e.L.glc_debug_attn_bench(e.h, 20, v, stamps if rep == 0 else 0, cs)
        if ms < 0:
            print("ERR", e.L.glc_last_error().decode()); continue
        print(f"{dtype} variant {v}: {ms:.4f} ms  {flops/ms/1e9:7.1f} TF (algorithmic)   checksum {cs[0]:.6e} {cs[1]:.6e}", flush=True)
e.close()
