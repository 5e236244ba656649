"""Quick on-GPU numerics sweep (developer tool): HIP engine vs C oracle on the golden cases."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights, synth
from gliclass.c_amd.engine import Engine
import oracle_c

def sig(x): return 1/(1+np.exp(-x.astype(np.float64)))
cases = sorted(glob.glob(os.path.join(ROOT, "tests/golden/*_b*_s*.npz")))
wc = {}
only = sys.argv[1:] 
for dt in ("f32", "f16", "bf16"):
    eng = {}
    for p in cases:
        g = np.load(p); cname = str(g["config"])
        if only and not any(o in p for o in only): continue
        if cname not in wc: wc[cname] = weights.make_weights(CONFIGS[cname], 42)
        cfg, w = CONFIGS[cname], wc[cname]
        if cname not in eng:
            t = time.time(); eng[cname] = Engine(cfg, w, dtype=dt); print(f"[{dt}] engine {cname} created in {time.time()-t:.1f}s", flush=True)
        e = eng[cname]
        ids, mask = g["ids"].astype(np.int64), g["mask"].astype(np.int64)
        e.keep_hidden(True)
        for impl in ((1,) if dt == "f32" else (1, 2)):
            e.set_attention_impl(impl)
            lg = e.forward(ids, mask)
            B, S = ids.shape
            pos = g["sample_pos"]; hs = g["hidden_samples"]
            errs = []
            for wl in range(cfg.layers + 1):
                h = e.hidden(wl, B, S)
                got = h[:, pos, :][..., :hs.shape[-1]][:, :hs.shape[2]]
                valid = mask[:, pos][:, :hs.shape[2]].astype(bool)
                errs.append(float(np.abs(got[valid] - hs[wl][valid]).max()))
            dl = np.abs(lg - g["logits"]).max(); dp = np.abs(sig(lg) - g["probs"]).max()
            e.keep_hidden(False)
            lgp = e.forward(ids, mask)   # pruned last layer path (default)
            e.keep_hidden(True)
            dpp = np.abs(sig(lgp) - g["probs"]).max()
            print(f"[{dt} attn={impl}] {os.path.basename(p):24s} pruned_prob_err {dpp:.2e} logit_err {dl:.2e} prob_err {dp:.2e} hidden_err {['%.1e'%x for x in errs]} finite={np.isfinite(lg).all()}", flush=True)
    for e in eng.values(): e.close()
