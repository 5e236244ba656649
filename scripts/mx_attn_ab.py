"""Developer tool (GPU): MX pipeline with attention on MX tiles vs on split-f16 units, interleaved same-process rounds (c3 shape)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
ids, mask, _ = synth.make_inputs(cfg, 64, 1024, 8, seed=1234)
e = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
e.set_length_buckets(1)
probs = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
out = {}
for rnd in range(3):
    for on in (0, 1):
        e.set_mx_attention(bool(on))
        out[on] = e.forward(ids, mask)
        e.profile(True)
        t0 = time.perf_counter()
        for _ in range(6): e.forward(ids, mask)
        dt = (time.perf_counter() - t0) / 6
        pr = e.profile_read(); e.profile(False)
        per = {k: v[0] / max(v[1], 1) for k, v in pr.items() if v[1]}
        print(f"r{rnd} mx_attention={on}: forward {dt*1e3:7.2f} ms | " + "  ".join(f"{k} {per[k]*1e3:.0f}us" for k in ("gemm_qkv", "attention", "gemm_attn_out", "gemm_ffn1_gelu", "gemm_ffn2", "last_layer_pruned") if k in per), flush=True)
print(f"MX-tile attention vs split-unit attention (both inside the MX pipeline): max |dp| {np.abs(probs(out[1]) - probs(out[0])).max():.3e}")
e.close()
