"""Developer tool (GPU, make DEV=1): two variants of the MX attention microbenchmark (glc_debug_attn_bench variant words) on the SAME MX tiles —
context rows compared element by element, then interleaved timing.  usage: attn_variant_ab.py VARIANT_A VARIANT_B   env: GLC_SHAPES, GLC_REPS"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
va, vb = int(sys.argv[1]), int(sys.argv[2])
cfg = CONFIGS[os.environ.get("GLC_CONFIG", "base")]
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("GLC_SHAPES", "64x1024,8x512,3x192,2x64,5x640,2x1536").split(",")]
reps = int(os.environ.get("GLC_REPS", 4))
e = Engine.from_spec(cfg, f"synthetic:{cfg.name}:42", dtype="f32")
e.set_length_buckets(1); e.set_group_split(2)
P = 2 * cfg.att_span
bad = 0
for (B, S) in shapes:
    for ragged in (False, True):
        ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3, ragged=ragged)
        e.L.glc_debug_set_stop(e.h, 1); e.forward(ids, mask)
        if not e.last_mx_attention(): continue
        Sp = (S + 63) // 64 * 64; rows = B * Sp; out = {}
        for v in (va, vb):
            cs = (ctypes.c_double * 2)()
            if e.L.glc_debug_attn_bench(e.h, 1, v, 0, cs) < 0: print("ERR", e.L.glc_last_error().decode()); sys.exit(1)
            buf = np.zeros((rows, cfg.hidden), np.float32)
            e.L.glc_debug_read_workspace(e.h, 2, rows, buf.ctypes.data_as(ctypes.c_void_p))
            out[v] = buf.reshape(B, Sp, cfg.hidden)
        valid = np.zeros((B, Sp), bool); valid[:, :S] = mask.astype(bool)
        same = np.array_equal(out[va][valid], out[vb][valid]); bad += 0 if same else 1
        print(f"B={B} S={S} ragged={ragged}: identical {same} (max |diff| {np.abs(out[va] - out[vb])[valid].max():.3e})", flush=True)
B, S = shapes[0]
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3)
e.L.glc_debug_set_stop(e.h, 1); e.forward(ids, mask)
flops = B * S * (4.0 * S * cfg.hidden + 4.0 * P * cfg.hidden)
for rep in range(reps):
    for v in (va, vb):
        cs = (ctypes.c_double * 2)()
        ms = e.L.glc_debug_attn_bench(e.h, 20, v, 0, cs)
        print(f"variant {v}: {ms:.4f} ms  {flops/ms/1e9:7.1f} TF (algorithmic)", flush=True)
e.L.glc_debug_set_stop(e.h, -1); e.close()
sys.exit(0 if bad == 0 else 2)
