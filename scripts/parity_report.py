"""Developer tool (GPU): measured parity of every committed fixture (tests/golden/*.npz, generated from HF DebertaV2Model /
Qwen2Model) per operand type: max |sigmoid(logit) - golden prob| and max hidden-state error on attended rows.
The tests assert bounds; this prints the actual values for docs/LOG_r01-r05.md §2."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
sig = lambda x: 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))
cases = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*_b*_s*.npz")))
wcache = {}
print(f"{'fixture':22s} {'config':9s} " + " ".join(f"{d+' prob':>11s} {d+' hid':>10s}" for d in ("f32", "f16", "bf16")))
for path in cases:
    g = np.load(path)
    cname = str(g["config"])
    cfg = CONFIGS[cname]
    if cname not in wcache:
        wcache[cname] = weights.make_weights(cfg, 42)
    ids, mask = g["ids"].astype(np.int64), g["mask"].astype(np.int64)
    B, S = ids.shape
    pos, hs = g["sample_pos"], g["hidden_samples"]
    valid = mask[:, pos][:, : hs.shape[2]].astype(bool)
    row = []
    for dt in ("f32", "f16", "bf16"):
        e = Engine(cfg, wcache[cname], dtype=dt)
        e.keep_hidden(True)
        lg = e.forward(ids, mask)
        perr = float(np.abs(sig(lg) - g["probs"]).max())
        herr = 0.0
        for which in range(cfg.layers + 1):
            got = e.hidden(which, B, S)[:, pos, :][..., : hs.shape[-1]][:, : hs.shape[2]]
            herr = max(herr, float(np.abs(got[valid] - hs[which][valid]).max()))
        e.close()
        row.append(f"{perr:11.2e} {herr:10.2e}")
    print(f"{os.path.basename(path)[:-4]:22s} {cname:9s} " + " ".join(row), flush=True)
