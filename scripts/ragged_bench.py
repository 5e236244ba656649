"""Developer tool (GPU): throughput of the host-buffer forward (glc_engine_forward, what run_inference calls) on a RAGGED batch
(lengths ~ U[S/2, S]) with and without length bucketing.  usage: ragged_bench.py [config=base] [B=64] [S=1024]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cname = sys.argv[1] if len(sys.argv) > 1 else "base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
cfg = CONFIGS[cname]
e = Engine.from_spec(cfg, f"synthetic:{cname}:42", dtype="f16")
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=1234, ragged=True)
full = np.ones_like(mask)
print(f"{cname} B={B} S={S}: mean length {mask.sum(1).mean():.0f}, min {mask.sum(1).min()}, max {mask.sum(1).max()}")
for name, m, groups in (("full-length rows", full, 1), ("ragged, one padded batch", mask, 1), ("ragged, length buckets (4)", mask, 4), ("ragged, length buckets (8)", mask, 8)):
    e.set_length_buckets(groups)
    for _ in range(2):
        out = e.forward(ids, m)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = e.forward(ids, m)
    dt = (time.perf_counter() - t0) / n
    print(f"{name:32s} {dt*1e3:8.2f} ms/batch  {B/dt:8.1f} seq/s  groups={e.L.glc_debug_last_forward_groups(e.h)}  finite={bool(np.isfinite(out).all())}")
e.close()
