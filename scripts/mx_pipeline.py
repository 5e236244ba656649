"""Developer tool (GPU): the MX cross-term pipeline (GLICLASS_MX) against the split-f16 default on the c3 shape, same process:
per-label probability error of all B x C probabilities (>= 3 weight seeds) and interleaved per-kernel / per-forward timing."""
import os, sys, time
os.environ["GLICLASS_MX"] = "build"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cname = os.environ.get("GLC_CONFIG", "base")
B, S = int(os.environ.get("GLC_B", 64)), int(os.environ.get("GLC_S", 1024))
cfg = CONFIGS[cname]
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=1234)
probs = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
worst, sq = 0.0, []
for seed in [int(x) for x in os.environ.get("GLC_SEEDS", "42,43,44").split(",")]:
    e = Engine.from_spec(cfg, f"synthetic:{cname}:{seed}", dtype="f32")
    e.set_length_buckets(1)
    ref = e.forward(ids, mask)
    assert e.last_group_split() and not e.last_mx()
    e.set_mx(True)
    got = e.forward(ids, mask)
    assert e.last_mx(), "the MX pipeline did not run"
    d = probs(got) - probs(ref)
    worst = max(worst, float(np.abs(d).max())); sq.append(float((d * d).mean()))
    print(f"seed {seed}: MX vs split-f16: max |dp| {np.abs(d).max():.3e}  rms {np.sqrt((d*d).mean()):.3e}  max |dlogit| {np.abs(got-ref).max():.3e}  finite {bool(np.isfinite(got).all())}", flush=True)
    if seed == 42:
        for rnd in range(3):
            for on in (0, 1):
                e.set_mx(bool(on))
                e.forward(ids, mask)
                e.profile(True)
                t0 = time.perf_counter()
                for _ in range(4): e.forward(ids, mask)
                dt = (time.perf_counter() - t0) / 4
                pr = e.profile_read(); e.profile(False)
                per = {k: v[0] / max(v[1], 1) for k, v in pr.items() if v[1]}
                tot = sum(v[0] for v in pr.values()) / 4
                print(f"  r{rnd} mx={on}: host-buffer forward {dt*1e3:7.2f} ms, kernels {tot:7.2f} ms | " + "  ".join(f"{k} {per[k]*1e3:.0f}us" for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn1_gelu", "gemm_ffn2", "attention", "layernorm", "last_layer_pruned") if k in per), flush=True)
    e.close()
print(f"worst max |dp| over seeds {worst:.3e}; rms over seeds {np.sqrt(np.mean(sq)):.3e}")
