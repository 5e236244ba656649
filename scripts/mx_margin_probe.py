"""Developer tool (GPU), VERDICT r4 item 6: where does the MX arithmetic lose its margin?  Replays the shape stream of scripts/soak_mx.py on the
base model, keeps the shapes with the largest |prob(MX) - prob(split)|, and splits each into its two sources: the projections on GX rows
(MX GEMMs with the attention on split units) and the attention on MX tiles (the difference).  usage: mx_margin_probe.py [n_shapes] [top]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth, weights
from gliclass.c_amd.engine import Engine
import oracle_c
n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 120
top = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = CONFIGS["base"]
e = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
rng = np.random.RandomState(20261004)
sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
rows = []
for it in range(n_shapes):
    B = int(rng.choice([8, 16, 24, 32, 48, 64])); S = int(rng.choice([256, 320, 512, 640, 768, 1024])); Cn = int(rng.randint(1, 9))
    seed = int(rng.randint(1 << 30)); ragged = bool(rng.randint(2)); e_b = int(rng.choice([1, 1, 4]))
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=seed, ragged=ragged)
    e.set_length_buckets(e_b)
    e.set_mx(True); e.set_mx_attention(True)
    a = e.forward(ids, mask)
    if not e.last_mx(): continue
    e.set_mx(False); x = e.forward(ids, mask); e.set_mx(True)
    rows.append((float(np.abs(sig(a) - sig(x)).max()), B, S, Cn, seed, ragged, e_b))
rows.sort(reverse=True)
print(f"{len(rows)} shapes on the MX pipeline; worst |prob(MX) - prob(split)| {rows[0][0]:.2e}, median {np.median([r[0] for r in rows]):.2e}")
w = weights.make_weights(cfg, 42)
for d, B, S, Cn, seed, ragged, e_b in rows[:top]:
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=seed, ragged=ragged)
    e.set_length_buckets(e_b)
    e.set_mx(True); e.set_mx_attention(True); full = e.forward(ids, mask)
    e.set_mx_attention(False); gemm_only = e.forward(ids, mask); e.set_mx_attention(True)
    e.set_mx(False); split = e.forward(ids, mask); e.set_mx(True)
    i = int(np.abs(sig(full) - sig(split)).max(axis=1).argmax())
    n = int(mask[i].sum())
    ref = oracle_c.forward(cfg, w, ids[i:i + 1, :n], mask[i:i + 1, :n])
    C = ref.shape[1]
    print(f"B={B} S={S} C={Cn} ragged={int(ragged)} buckets={e_b}: MX vs split {d:.2e} | MX GEMMs + split attention vs split {np.abs(sig(gemm_only) - sig(split)).max():.2e} | "
          f"worst row vs the oracle: MX {np.abs(sig(full[i:i+1, :C]) - sig(ref)).max():.2e}, split {np.abs(sig(split[i:i+1, :C]) - sig(ref)).max():.2e}; largest |logit| {np.abs(ref).max():.1f}", flush=True)
e.close()
