"""Developer tool (GPU): the MX cross-term GEMMs (GLC_MX mask) against the group-split split-f16 pipeline and the oracle.
usage: mx_check.py [config] [B] [S]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cname = sys.argv[1] if len(sys.argv) > 1 else "small"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cfg = CONFIGS[cname]
sig = lambda x: 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=11, ragged=True)
ref = None
if B * S <= 4096 and cname in ("mini", "small"):
    import oracle_c
    from gliclass.c_amd import weights
    w = weights.make_weights(cfg, 42)
    ref = oracle_c.forward(cfg, w, ids, mask)
outs = {}
for mx in (0, 1, 2, 3):
    os.environ["GLC_MX"] = str(mx)
    e = Engine.from_spec(cfg, f"synthetic:{cfg.name}:42", dtype="f32")
    e.set_group_split(2)
    e.set_length_buckets(1)
    o = e.forward(ids, mask)
    assert e.last_group_split()
    t0 = time.time()
    for _ in range(3): e.forward(ids, mask)
    dt = (time.time() - t0) / 3
    outs[mx] = o
    line = f"GLC_MX={mx}: {dt*1e3:8.2f} ms/forward  finite={np.isfinite(o).all()}"
    if mx: line += f"  max|dprob| vs split-f16 {np.abs(sig(o) - sig(outs[0])).max():.3e}"
    if ref is not None: line += f"  vs oracle {np.abs(sig(o) - sig(ref)).max():.3e}"
    print(line, flush=True)
    e.close()
