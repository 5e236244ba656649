"""Developer tool (GPU): the bucket-space MX attention (attention_mx2.hip) against the band kernel (attention_mx.hip) on the SAME MX tiles —  [needs a developer build: make -C gliclass/c_amd DEV=1 (csrc/dev/attention_mx2.hip)]
context rows compared element by element (both kernels run the same products; differences are accumulation order), then interleaved timing.
usage: attn_mx2_check.py [stamps 0/1]   env: GLC_B, GLC_S, GLC_CONFIG, GLC_RAGGED"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
stamps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = CONFIGS[os.environ.get("GLC_CONFIG", "base")]
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("GLC_SHAPES", "64x1024").split(",")]
e = Engine.from_spec(cfg, f"synthetic:{cfg.name}:42", dtype="f32")
e.set_length_buckets(1)
P = 2 * cfg.att_span
worst = 0.0
for (B, S) in shapes:
    for ragged in (False, True):
        ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3, ragged=ragged)
        e.L.glc_debug_set_stop(e.h, 1)         # leave the forward after layer 0's attention: Q / K / V^T hold that layer's MX tiles
        e.forward(ids, mask)
        if not e.last_mx_attention():
            print(f"B={B} S={S}: the forward did not take the MX attention, skipped"); continue
        Sp = (S + 63) // 64 * 64
        rows = B * Sp
        out = {}
        for v in (128, 128 | 8192, 128 | 16384):
            cs = (ctypes.c_double * 2)()
            ms = e.L.glc_debug_attn_bench(e.h, 2, v, 0, cs)
            if ms < 0:
                print("ERR", e.L.glc_last_error().decode()); sys.exit(1)
            buf = np.zeros((rows, cfg.hidden), np.float32)
            if e.L.glc_debug_read_workspace(e.h, 2, rows, buf.ctypes.data_as(ctypes.c_void_p)) != 0:
                print("ERR", e.L.glc_last_error().decode()); sys.exit(1)
            out[v] = buf.reshape(B, Sp, cfg.hidden)
        a, b = out[128], out[128 | 8192]
        print("  band kernel with recomputed gather addresses identical to the spilling build:", np.array_equal(out[128], out[128 | 16384]))
        valid = np.zeros((B, Sp), bool); valid[:, :S] = mask.astype(bool)      # rows whose query is attended (others are don't-care)
        d = np.abs(a - b)[valid]
        scale = np.abs(a[valid]).max()
        worst = max(worst, d.max() / scale)
        print(f"B={B} S={S} ragged={ragged}: max |ctx(mx2) - ctx(band)| = {d.max():.3e} (rows max {scale:.3f}), rms {np.sqrt((d**2).mean()):.3e}, finite {np.isfinite(b[valid]).all()}", flush=True)
print("worst relative difference", worst)
B, S = shapes[0]
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3)
e.L.glc_debug_set_stop(e.h, 1)
e.forward(ids, mask)
flops = B * S * (4.0 * S * cfg.hidden + 4.0 * P * cfg.hidden)
for rep in range(3):
    for v in [128, 128 | 8192] + [128 | int(x) for x in os.environ.get("GLC_BANDV", "").split(",") if x] + [128 | 8192 | int(x) for x in os.environ.get('GLC_ABL', '').split(',') if x]:
        cs = (ctypes.c_double * 2)()
        ms = e.L.glc_debug_attn_bench(e.h, 20, v, stamps if rep == 0 else 0, cs)
        print(f"variant {v}: {ms:.4f} ms  {flops/ms/1e9:7.1f} TF (algorithmic)", flush=True)
e.L.glc_debug_set_stop(e.h, -1)
e.close()
sys.exit(0 if worst < 1e-4 else 2)
