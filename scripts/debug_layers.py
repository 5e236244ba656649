#!/usr/bin/env python3
"""Per-layer max |hidden - oracle| of one tiny forward (debug aid): debug_layers.py [dtype] [config]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gliclass.c_amd import synth, weights
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd.engine import Engine
import oracle_c
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "tiny"]
w = weights.make_weights(cfg, 42)
ids, mask, _ = synth.make_inputs(cfg, 2, 128, 3, seed=7, ragged=False)
ref, hid = oracle_c.forward(cfg, w, ids, mask, want_hidden=True)[:2]
eng = Engine(cfg, w, dtype=dtype)
eng.keep_hidden(True)
got = eng.forward(ids, mask)
print("logits max err", np.abs(got - ref).max())
for l in range(cfg.layers + 1):
    h = eng.hidden(l, 2, 128)
    print("hidden", l, "max err", np.abs(h - hid[l]).max(), "ref absmax", np.abs(hid[l]).max())
