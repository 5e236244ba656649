"""Developer tool (GPU): replays the case stream of scripts/gpu_soak.py for one model in the default mode and prints the worst cases of the
forced group-split pipeline next to the plain-fp32-row pipeline on the same inputs.  usage: soak_worst_gpu.py [seed] [n_cases] [model]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle_c
from gliclass.c_amd import synth, weights
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd.engine import Engine
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 21
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cname = sys.argv[3] if len(sys.argv) > 3 else "small"
rng = np.random.default_rng(seed)
sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
cfg = CONFIGS[cname]
w = weights.make_weights(cfg, 42)
eng = Engine(cfg, w, dtype="f32")
rows = []
for case in range(ncase):
    _ = int(rng.integers(0, 1))           # (model pick of the soak: one model here)
    B = int(rng.integers(1, 13))
    S = int(rng.integers(1, 1401)) if rng.random() < 0.8 else int(rng.integers(1, 48))
    cmax = int(rng.integers(0, 9))
    lpr = [int(x) for x in rng.integers(0, cmax + 1, size=B)]
    S = max(S, 2 + 3 * max(lpr + [0]) + 2)
    ids, mask, _ = synth.make_inputs(cfg, B, S, max(cmax, 1), seed=int(rng.integers(0, 1 << 30)), ragged=bool(rng.integers(0, 2)), labels_per_row=lpr)
    if rng.random() < 0.25 and S > 20:
        n0 = int(mask[0].sum()); lo = max(n0 // 2, 2 + 3 * lpr[0] + 1)
        if n0 > 16 and lo + 3 < n0: mask[0, lo: lo + 3] = 0
    ref = oracle_c.forward(cfg, w, ids, mask)
    if not ref.size: continue
    out = {}
    for name, gsm, lnf in (("plain", 0, True), ("gs", 2, False), ("gs+lnf", 2, True)):
        eng.set_group_split(gsm); eng.set_ln_fused(lnf)
        got = eng.forward(ids, mask)
        out[name] = float(np.abs(sig(got) - sig(ref)).max())
    rows.append((out["gs+lnf"], case, B, S, lpr, out))
rows.sort(reverse=True)
for r in rows[:6]:
    print(f"case {r[1]:3d} B={r[2]:2d} S={r[3]:4d} labels={r[4]}  plain {r[5]['plain']:.2e}  gs {r[5]['gs']:.2e}  gs+lnf {r[5]['gs+lnf']:.2e}", flush=True)
eng.close()
