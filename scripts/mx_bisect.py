"""Developer tool (GPU): stage-by-stage comparison of the MX pipeline with the group-split pipeline (workspace rows decoded to fp32)."""
import os, sys
os.environ["GLICLASS_MX"] = "build"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth, weights
from gliclass.c_amd.engine import Engine
cname = os.environ.get("GLC_CONFIG", "mini")
cfg = CONFIGS[cname]
B, S = int(os.environ.get("GLC_B", 4)), int(os.environ.get("GLC_S", 192))
w = weights.make_weights(cfg, 42)
ids, mask, _ = synth.make_inputs(cfg, B, S, 3, seed=5)
e = Engine(cfg, w, dtype="f32")
e.set_length_buckets(1); e.set_group_split(2)
# GLC_BISECT_ATTN=1: compare MX attention (on) against the MX pipeline with split-unit attention; else MX pipeline vs group-split pipeline
ATT = bool(os.environ.get("GLC_BISECT_ATTN"))
rows = B * S
NAMES = {0: "X", 1: "H1", 2: "CTX", 3: "FF", 4: "T1", 5: "statsA", 6: "statsB", 7: "Q units", 8: "K units", 9: "V^T units"}
def read(which):
    W = cfg.inter if which == 3 else (2 if which in (5, 6) else cfg.hidden)
    out = np.zeros((rows, W), np.float32)
    assert e.L.glc_debug_read_workspace(e.h, which, rows, out.ctypes.data) == 0, e.L.glc_last_error().decode()
    if which >= 7:      # split-f16 units [8 hi | 8 lo] halves -> values
        hv = out.view(np.float16).reshape(-1, 2, 8).astype(np.float32)
        return (hv[:, 0] + hv[:, 1]).reshape(rows, -1)
    return out
for layer in range(cfg.layers - 1):
    for k, bufs in ((0, [0] if ATT else [0, 7, 8, 9]), (1, [2]), (2, [1]), (3, [3]), (4, [0, 4])):
        snap = {}
        for on in (0, 1):
            if ATT: e.set_mx(True); e.set_mx_attention(bool(on))
            else: e.set_mx(bool(on)); e.set_mx_attention(False)
            e.L.glc_debug_set_stop(e.h, 10 * layer + k)
            e.forward(ids, mask)
            assert e.last_group_split() and e.last_mx() == (True if ATT else bool(on))
            snap[on] = {b: read(b) for b in bufs + ([6] if k == 2 else []) + ([5] if k == 4 else [])}
        for b in snap[0]:
            a0, a1 = snap[0][b], snap[1][b]
            d = np.abs(a1 - a0)
            print(f"layer {layer} stage {k} {NAMES[b]:6s}: max |mx - gs| {d.max():.3e} (max |gs| {np.abs(a0).max():.3e}, rms gs {np.sqrt((a0.astype(np.float64)**2).mean()):.3e}) worst row {int(d.max(1).argmax())}", flush=True)
e.L.glc_debug_set_stop(e.h, -1)
e.close()
