"""Developer probe (GPU): two engines (two streams) in ONE process, each running half the c3 batch (B = 32) concurrently, against one engine on
the full batch (B = 64).  The projections run at the chip's power envelope, the attention does not: do interleaved kernels use the headroom?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
S, Cn, steps = 1024, 8, 20
def setup(B, seed):
    e = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=seed, ragged=False)
    d_ids, d_mask, d_out = e.dev_alloc(ids.nbytes), e.dev_alloc(mask.nbytes), e.dev_alloc(B * Cn * 4)
    e.h2d(d_ids, ids); e.h2d(d_mask, mask)
    for _ in range(3): e.forward_device(d_ids, d_mask, B, S, Cn, d_out)
    e.sync()
    return e, d_ids, d_mask, d_out, B
def run(ctx, n):
    e, d_ids, d_mask, d_out, B = ctx
    for _ in range(n): e.forward_device(d_ids, d_mask, B, S, Cn, d_out)
    e.sync()
full = setup(64, 1)
t0 = time.time(); run(full, steps); t1 = time.time()
print(f"one engine, B = 64: {64 * steps / (t1 - t0):8.1f} seq/s ({(t1 - t0) / steps * 1e3:.2f} ms per step)", flush=True)
a, b = setup(32, 2), setup(32, 3)
t0 = time.time(); run(a, steps); t1 = time.time()
print(f"one engine, B = 32: {32 * steps / (t1 - t0):8.1f} seq/s ({(t1 - t0) / steps * 1e3:.2f} ms per step)", flush=True)
ths = [threading.Thread(target=run, args=(c, steps)) for c in (a, b)]
t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]; t1 = time.time()
print(f"two engines, B = 32 each, concurrent streams: {64 * steps / (t1 - t0):8.1f} seq/s ({(t1 - t0) / steps * 1e3:.2f} ms per pair of steps)", flush=True)
