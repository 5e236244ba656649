import os, sys, numpy as np, time
sys.path.insert(0, '/root/repo')
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
sig = lambda x: 1/(1+np.exp(-x.astype(np.float64)))
ids, mask, _ = synth.make_inputs(cfg, 64, 1024, 8, seed=1234)
worst = 0; rmss = []
for seed in (42, 43, 44):
    e = Engine.from_spec(cfg, f"synthetic:base:{seed}", dtype="f32"); e.set_length_buckets(1)
    got = e.forward(ids, mask)
    e.set_mx(False); ref = e.forward(ids, mask); e.set_mx(True)
    d = sig(got) - sig(ref); worst = max(worst, float(np.abs(d).max())); rmss.append(float(np.sqrt((d*d).mean())))
    if seed == 42:
        t0 = time.perf_counter()
        for _ in range(10): e.forward(ids, mask)
        ms = (time.perf_counter() - t0) * 100
    e.close()
print(f"GLC_ATTN_PV16={os.environ.get('GLC_ATTN_PV16','0')}: 3 seeds x 512 probabilities vs split: max {worst:.2e}, rms {np.sqrt(np.mean(np.square(rmss))):.2e}; forward (host buffers) {ms:.2f} ms")
