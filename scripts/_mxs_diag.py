import ctypes, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS["base"]
e = Engine.from_spec(cfg, f"synthetic:{cfg.name}:42", dtype="f32")
e.set_length_buckets(1); e.set_group_split(2)
for (B, S) in [(1, 192), (1, 320), (1, 1024)]:
    ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3, ragged=False)
    e.L.glc_debug_set_stop(e.h, 1); e.forward(ids, mask)
    Sp = (S + 63) // 64 * 64; rows = B * Sp; out = {}
    for v in (128, 128 | 32768):
        cs = (ctypes.c_double * 2)()
        e.L.glc_debug_attn_bench(e.h, 1, v, 0, cs)
        buf = np.zeros((rows, cfg.hidden), np.float32)
        e.L.glc_debug_read_workspace(e.h, 2, rows, buf.ctypes.data_as(ctypes.c_void_p))
        out[v] = buf.reshape(B, Sp, cfg.heads, 64)
    d = np.abs(out[128] - out[128 | 32768])[0]          # [Sp, nh, 64]
    per_tile = d.reshape(Sp // 32, 32, cfg.heads, 64).max(axis=(1, 3))     # [tiles, heads]
    print(f"S={S}: max diff per 32-query tile (rows) x head (cols 0..3):")
    for t in range(Sp // 32): print(f"  tile {t:2d}: " + " ".join(f"{x:8.1e}" for x in per_tile[t, :4]))
e.close()
