"""Developer tool (GPU): soak of the default (MX) pipeline — random batch shapes on the base model for a few minutes, every forward run
twice (bit-identical logits expected: same shapes, same kernels, fixed accumulation orders; a synchronisation hazard in the LDS rings /
images of gemm256x.hip or attention_mx.hip would show as a mismatch) and, every few rounds, rows checked against the three-MFMA arithmetic
(GLICLASS_MX=0) within the test tolerance.  usage: soak_mx.py [seconds] [config]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
cname = sys.argv[2] if len(sys.argv) > 2 else "base"
cfg = CONFIGS[cname]
e = Engine.from_spec(cfg, f"synthetic:{cname}:42", dtype="f32")
if os.environ.get("GLC_SOAK_MX2"):          # the bucket-space attention kernel (attention_mx2.hip) instead of the band kernel
    e.set_mx2(True); print("attention: attention_mx2.hip")
rng = np.random.RandomState(20261004)
sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
t0 = time.time(); rounds = 0; mism = 0; worst = 0.0; mx_rounds = 0
while time.time() - t0 < budget:
    B = int(rng.choice([8, 16, 24, 32, 48, 64])); S = int(rng.choice([256, 320, 512, 640, 768, 1024])); Cn = int(rng.randint(1, 9))
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=int(rng.randint(1 << 30)), ragged=bool(rng.randint(2)))
    e_b = int(rng.choice([1, 1, 4]))
    e.set_length_buckets(e_b)
    a = e.forward(ids, mask); mx = e.last_mx()
    b = e.forward(ids, mask)
    rounds += 1; mx_rounds += int(mx)
    if not np.array_equal(a, b):
        mism += 1
        print(f"MISMATCH B={B} S={S} C={Cn} mx={mx}: max |d logit| {np.abs(a - b).max():.3e}", flush=True)
    if rounds % 2 == 0 and mx:
        e.set_mx(False); x = e.forward(ids, mask); e.set_mx(True)
        d = float(np.abs(sig(a) - sig(x)).max())
        if d > worst: worst = d; print(f"new worst MX vs split {d:.3e} at B={B} S={S} C={Cn} ragged={int(mask.min() == 0)} buckets={e_b}", flush=True)
        if d > 5e-4: print(f"TOLERANCE B={B} S={S} C={Cn}: MX vs split {d:.3e}", flush=True)
    if rounds % 20 == 0: print(f"{time.time() - t0:6.0f} s: {rounds} shapes ({mx_rounds} on the MX pipeline), {mism} mismatches, worst MX vs split {worst:.2e}", flush=True)
print(f"done: {rounds} shapes ({mx_rounds} MX), {mism} run-to-run mismatches, worst MX vs split {worst:.2e}")
e.close()
sys.exit(1 if mism else 0)
