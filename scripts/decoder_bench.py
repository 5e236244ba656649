"""Developer tool (GPU): BASELINE.json configs[4] — decoder backbone at its full size (qwen-1.5b shape, B=16, S=2048, 8 labels,
synthetic weights), timing + per-kernel profile.  Not the headline bench (bench.py is config c3); numbers go to DESIGN.md.
usage: decoder_bench.py [dtype=bf16] [config=qwen-1.5b] [B=16] [S=2048]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cname = sys.argv[2] if len(sys.argv) > 2 else "qwen-1.5b"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
S = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
Cn = 8
cfg = CONFIGS[cname]
t0 = time.time()
e = Engine.from_spec(cfg, f"synthetic:{cname}:42", dtype=dtype)
print(f"engine ready in {time.time() - t0:.1f} s", flush=True)
ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234, ragged=False)
d_ids, d_mask, d_log = e.dev_alloc(ids.nbytes), e.dev_alloc(mask.nbytes), e.dev_alloc(B * Cn * 4)
e.h2d(d_ids, ids.astype(np.int64)); e.h2d(d_mask, mask.astype(np.int64))
for _ in range(2):
    e.forward_device(d_ids, d_mask, B, S, Cn, d_log)
e.sync()
steps = 5
e.timer_start()
for _ in range(steps):
    e.forward_device(d_ids, d_mask, B, S, Cn, d_log)
ms = e.timer_stop_ms() / steps
logits = np.zeros((B, Cn), np.float32)
e.d2h(logits, d_log)
e.profile(True)                      # the profile holds the events of the LAST forward
e.forward_device(d_ids, d_mask, B, S, Cn, d_log)
e.sync()
prof = e.profile_read()
fl = cfg.flops_per_seq(S, Cn) * B
out = dict(config=cname, dtype=dtype, B=B, S=S, ms_per_batch=round(ms, 3), seq_per_s=round(B / ms * 1e3, 2), tflops=round(fl / ms / 1e9, 1),
           frac_of_2500=round(fl / ms / 1e9 / 2500, 4), finite=bool(np.isfinite(logits).all()), logits_row0=[round(float(x), 4) for x in logits[0]],
           per_kernel={k: dict(ms_per_fwd=round(v[0], 3), launches=v[1]) for k, v in prof.items() if v[1]})
print(json.dumps(out))
e.close()
