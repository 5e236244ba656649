"""Developer tool (GPU): forward time of small shapes (the reference's own operating point: batches of 8 short texts,
/root/reference/include/configs.h:4-7) with the small-M GEMM dispatch on / off.  Run with GLC_GEMM_SMALL_M=0 for the A/B."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
for cname, B, S, Cn in (("small", 1, 128, 4), ("small", 8, 128, 4), ("small", 8, 512, 8), ("base", 8, 256, 8), ("base", 8, 1024, 8), ("base", 16, 1024, 8), ("base", 64, 1024, 8)):
    cfg = CONFIGS[cname]
    e = Engine.from_spec(cfg, f"synthetic:{cname}:42", dtype="f16")
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=3, ragged=False)
    d_ids, d_mask, d_log = e.dev_alloc(ids.nbytes), e.dev_alloc(mask.nbytes), e.dev_alloc(B * Cn * 4)
    e.h2d(d_ids, ids); e.h2d(d_mask, mask)
    for _ in range(5): e.forward_device(d_ids, d_mask, B, S, Cn, d_log)
    e.sync()
    n = 100 if B * S < 16384 else 10
    e.timer_start()
    for _ in range(n): e.forward_device(d_ids, d_mask, B, S, Cn, d_log)
    ms = e.timer_stop_ms() / n
    lg = np.zeros((B, Cn), np.float32); e.d2h(lg, d_log)
    print(f"{cname:6s} B={B:3d} S={S:5d}: {ms:8.3f} ms/fwd  {B/ms*1e3:9.1f} seq/s  logit0 {lg[0,0]:+.4f}", flush=True)
    e.close()
