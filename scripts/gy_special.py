import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import weights
from gliclass.c_amd.engine import Engine
e = Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f16")
out = (C.c_double * 5)()
for spec in (1, 2, 3, 4):
    for fmt in (0, 10):
        rc = e.L.glc_debug_gemm_mx_check(e.h, 256, 256, 64, C.c_float(-float(spec)), C.c_float(1.0), 0 + fmt, out)
e.close()
