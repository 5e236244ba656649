"""Developer tool (GPU): where engine creation time goes (weight source vs engine_create)."""
import sys, time, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd import _lib
from gliclass.c_amd.engine import DTYPES
for name in (sys.argv[1:] or ["base"]):
    M = _lib.model(); L = _lib.hip()
    for rep in range(2):
        w = _lib.Weights()
        t0 = time.time()
        assert M.glc_weights_load(f"synthetic:{name}:42".encode(), C.byref(w)) == 0
        t1 = time.time()
        h = L.glc_engine_create(C.byref(w.cfg), C.cast(w.tensors, C.POINTER(C.c_void_p)), w.n_tensors, 0, DTYPES["f32"])
        t2 = time.time()
        M.glc_weights_free(C.byref(w))
        L.glc_engine_destroy(h)
        print(name, f"weights_load {t1 - t0:.2f} s   engine_create {t2 - t1:.2f} s", flush=True)
