"""Developer tool (GPU, under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE): launches of the MX attention kernels only — band kernel with 4-wave and
8-wave workgroups, bucket-space kernel — on the c3 shape, for per-kernel HBM byte counts (WRITE_SIZE 346 MB against 201 MB algorithmic)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
e = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
e.set_length_buckets(1)
ids, mask, _ = synth.make_inputs(cfg, 64, 1024, 8, seed=3)
e.L.glc_debug_set_stop(e.h, 1)
e.forward(ids, mask)
for v in (128 | 1024, 128 | 2048, 128 | 8192):
    cs = (ctypes.c_double * 2)()
    ms = e.L.glc_debug_attn_bench(e.h, 5, v, 0, cs)
    print("variant", v, ms, flush=True)
e.L.glc_debug_set_stop(e.h, -1)
e.close()
