"""Developer tool (GPU, make DEV=1): time and stamp one MX attention variant at c3 (glc_debug_attn_bench).  usage: attn_stamps.py VARIANT [VARIANT ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
e = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
e.set_length_buckets(1); e.set_group_split(2)
ids, mask, _ = synth.make_inputs(cfg, 64, 1024, 8, seed=3)
e.L.glc_debug_set_stop(e.h, 1); e.forward(ids, mask)
cs = (ctypes.c_double * 2)()
for rep in range(2):
    for v in sys.argv[1:]:
        print("variant", v, "ms", e.L.glc_debug_attn_bench(e.h, 20, int(v), 0, cs), flush=True)
for v in sys.argv[1:]:
    print("stamped", v, e.L.glc_debug_attn_bench(e.h, 3, int(v), 1, cs), flush=True)
e.L.glc_debug_set_stop(e.h, -1); e.close()
