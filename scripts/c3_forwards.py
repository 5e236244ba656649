"""Developer tool (GPU): a few c3 forwards under the current GLC_MX, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
ids, mask, _ = synth.make_inputs(cfg, 64, 1024, 8, seed=11)
e = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
e.set_length_buckets(1)
for _ in range(4): e.forward(ids, mask)
e.close()
