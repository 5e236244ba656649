"""Developer probe (GPU): is the attention kernel held by the power envelope as the projections are (profiles/r04/gemm_power_envelope.txt)?
The MX band kernel on the c3 shape with the model's Q / K / V and position tables, and with all of them zero (query / key / value projections
zeroed: same launches, same traffic, no toggling operands)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth, weights
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
ids, mask, _ = synth.make_inputs(cfg, 64, 1024, 8, seed=3, ragged=False)
for tag in ("model operands", "zero operands"):
    w = dict(weights.make_weights(cfg, 42))
    if tag.startswith("zero"):
        n = 0
        for k in list(w):
            if any(t in k for t in ("query_proj", "key_proj", "value_proj")):
                w[k] = np.zeros_like(w[k]); n += 1
        assert n >= 3 * cfg.layers, n
    e = Engine(cfg, w, dtype="f32")
    e.set_length_buckets(1)
    e.forward(ids, mask)
    assert e.last_mx() and e.last_mx_attention()
    for rep in range(3):
        cs = (ctypes.c_double * 2)()
        ms = e.L.glc_debug_attn_bench(e.h, 20, 128, 0, cs)
        print(f"{tag}: attn_mx_kernel {ms:.4f} ms per launch   checksum {cs[0]:.4e}", flush=True)
    e.close()
