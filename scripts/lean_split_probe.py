"""Developer tool (GPU): the per-wave band kernel with split fragments (impl 2) against the oracle; GLC_ATTN_SPLIT_LEAN=1 picks the lean loop
(docs/LOG_r01-r05.md section 2: the instantiation hipcc miscompiles when its SLP vectoriser is on)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_c
from gliclass.c_amd import synth, weights
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["mini"]; w = weights.make_weights(cfg, 42)
e = Engine(cfg, w, dtype="f32")
e.L.glc_debug_set_attention_impl(e.h, 2)
sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
for (B, S, seed) in ((2, 96, 1), (2, 300, 2), (2, 700, 3), (1, 1300, 4)):
    ids, mask, _ = synth.make_inputs(cfg, B, S, 3, seed=seed, ragged=True)
    ref = oracle_c.forward(cfg, w, ids, mask)
    got = e.forward(ids, mask)
    print(B, S, "err", float(np.abs(sig(got) - sig(ref)).max()), flush=True)
