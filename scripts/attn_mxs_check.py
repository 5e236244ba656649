"""Developer tool (GPU, make DEV=1 — scripts/dev_variant_build.sh + GLC_HIP_SO): the role-split MX attention (csrc/dev/attention_mxs.hip) against the band kernel (attention_mx.hip) on the SAME MX tiles —
the context rows must agree to rounding (same products; only the leaving p2c block is summed in another order) — then interleaved timing on the first shape.
usage: attn_mxs_check.py [stamps 0/1]   env: GLC_SHAPES (BxS,...), GLC_CONFIG, GLC_REPS"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd.config import CONFIGS
from gliclass.c_amd import synth
from gliclass.c_amd.engine import Engine
cfg = CONFIGS[os.environ.get("GLC_CONFIG", "base")]
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("GLC_SHAPES", "64x1024,8x512,3x192,2x64,5x640,2x1536").split(",")]
reps = int(os.environ.get("GLC_REPS", 4))
stamps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
e = Engine.from_spec(cfg, f"synthetic:{cfg.name}:42", dtype="f32")
e.set_length_buckets(1)
e.set_group_split(2)
P = 2 * cfg.att_span
bad = 0
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
        for v in (128, 128 | 32768):
            cs = (ctypes.c_double * 2)()
            ms = e.L.glc_debug_attn_bench(e.h, 1, v, 0, cs)
            if ms < 0:
                print("ERR", e.L.glc_last_error().decode()); sys.exit(1)
            buf = np.zeros((rows, cfg.hidden), np.float32)
            if e.L.glc_debug_read_workspace(e.h, 2, rows, buf.ctypes.data_as(ctypes.c_void_p)) != 0:
                print("ERR", e.L.glc_last_error().decode()); sys.exit(1)
            out[v] = buf.reshape(B, Sp, cfg.hidden)
        a, b = out[128], out[128 | 32768]
        valid = np.zeros((B, Sp), bool); valid[:, :S] = mask.astype(bool)      # rows whose query is attended (others are don't-care)
        same = np.array_equal(a[valid], b[valid])
        d = np.abs(a - b)[valid]
        nz = int((d != 0).sum())
        rel = d.max() / max(np.abs(a[valid]).max(), 1e-30)
        # the same products in the same order except the leaving p2c block (16 x 16 MFMA shapes sum their k-steps in another order): <= ~2e-5 of the rows' scale
        ok = bool(np.isfinite(b[valid]).all()) and rel <= 5e-5
        bad += 0 if ok else 1
        print(f"B={B} S={S} ragged={ragged}: identical {same}  (differing elements {nz} of {d.size}, max |diff| {d.max():.3e} = {rel:.2e} of the rows' max; {'OK' if ok else 'FAIL'})", flush=True)
B, S = shapes[0]
ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3)
e.L.glc_debug_set_stop(e.h, 1)
e.forward(ids, mask)
flops = B * S * (4.0 * S * cfg.hidden + 4.0 * P * cfg.hidden)
for rep in range(reps):
    for v in [128, 128 | 32768] + [128 | 32768 | int(x) for x in os.environ.get('GLC_XV', '').split(',') if x]:
        cs = (ctypes.c_double * 2)()
        ms = e.L.glc_debug_attn_bench(e.h, 20, v, stamps if (rep == 0 and (v & 3) == 0) else 0, cs)
        print(f"variant {v}: {ms:.4f} ms  {flops/ms/1e9:7.1f} TF (algorithmic)", flush=True)
e.L.glc_debug_set_stop(e.h, -1)
e.close()
sys.exit(0 if bad == 0 else 2)
