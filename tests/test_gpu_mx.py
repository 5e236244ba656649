"""GPU (-m gpu): the MX cross-term GEMM (gemm256x.hip: a_hi*w_hi in f16 MFMAs + both cross terms in one block-scaled fp8 MFMA, GX rows)
against the split-f16 GEMM (gemm256s.hip, three f16 MFMAs per product, group-split rows) on the same random fp32 operands — every
epilogue path the pipeline uses, through the C-ABI's developer entry glc_debug_gemm_mx_check.  The product error bound of the MX
arithmetic is ~2^-15 relative (cross terms to ~4 bits); measured 1e-5 relative rms.  The whole-forward error of the MX pipeline is
asserted in test_gpu_parity.py (forced pipeline, mini / small) and test_gpu_fullsize.py (c3, three seeds; c4's shard)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODES = {0: "bias, plain fp32 out", 1: "GELU + LayerNorm fold, row out", 2: "residual (raw rows, LayerNorm on the fly), raw rows + partials out",
         3: "residual, plain fp32 out", 4: "QKV + LayerNorm fold, split-f16 units"}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_mx_gemm_every_epilogue_vs_split_gemm(mode, weights_for):
    from gliclass.c_amd.engine import Engine
    cfg, w = weights_for("tiny")
    eng = Engine(cfg, w, dtype="f16")
    out = (C.c_double * 5)()
    try:
        # (the last three shapes have N K >= 768 x 3072: bias / GELU / residual launches take the one-wave-per-SIMD 128 x 128 tile there (round 6) — K = 32 is its
        #  shortest loop: prologue + a peeled tail only — the others the 8-wave tile)
        for (M, N, K, a_amp, w_amp) in ((256, 256, 32, 1.0, 0.05), (512, 768, 768, 2.0, 0.05), (1024, 768, 3072, 1.0, 0.1), (256, 768, 96, 50.0, 1.0),
                                        (512, 3072, 768, 1.0, 0.05), (512, 1536, 1536, 2.0, 0.05), (256, 73728, 32, 1.0, 0.05), (256, 36864, 64, 1.0, 0.05)):
            if mode == 4:
                N = 768                      # N = 3 H, H = 256: a tile never straddles Q | K | V
            rc = eng.L.glc_debug_gemm_mx_check(eng.h, M, N, K, a_amp, w_amp, mode, out)
            assert rc == 0, eng.L.glc_last_error().decode()
            max_d, max_ref, rms_d, rms_ref, part_d = (out[i] for i in range(5))
            assert np.isfinite(max_d) and rms_ref > 0
            assert rms_d <= 4e-5 * rms_ref, (MODES[mode], M, N, K, rms_d / rms_ref)       # ~2^-15 per product; a layout error is O(1)
            assert max_d <= 4e-4 * max_ref, (MODES[mode], M, N, K, max_d / max_ref)
            if mode == 2:
                assert part_d <= 2e-3 * max(1.0, a_amp), (M, N, K, part_d)              # the LayerNorm partials follow the rows
    finally:
        eng.close()


def test_mx_pipeline_odd_batch_shapes():
    """The MX pipeline (GX projections, MX-tile attention with 4-wave workgroups) on batch shapes the headline does not have: sequence
    lengths that are no multiple of the workgroup's 128 queries (partly idle workgroups), one query tile per row, long rows, ragged rows
    — all probabilities against the three-MFMA arithmetic of the same engine (itself <= 2e-5 from the oracle, test_gpu_fullsize.py)."""
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["base"]
    eng = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
    sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    try:
        eng.set_length_buckets(1)
        # (64, 320, ragged) is the worst shape the round-3 soak found (2.1e-4, scripts/soak_mx.py)
        for (B, S, Cn, ragged) in ((300, 192, 3, True), (100, 320, 8, True), (64, 320, 8, True), (33, 1000, 5, True), (70, 448, 1, False), (17, 2048, 8, True), (1024, 64, 1, False)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=B + S, ragged=ragged)
            got = eng.forward(ids, mask)
            assert eng.last_mx() and np.isfinite(got).all(), (B, S)
            eng.set_mx(False)
            ref = eng.forward(ids, mask)
            eng.set_mx(True)
            assert not np.array_equal(got, ref)
            d = float(np.abs(sig(got) - sig(ref)).max())
            assert d <= 5e-4, (B, S, Cn, d)                    # (TOL_MX of test_gpu_parity.py: 2x the observed worst case)
        assert eng.L.glc_debug_mx_weight_bytes(eng.h) > 11 * 4 * 7077888       # (base: 12 layers x ~7.1 M projection weights x 4 bytes, minus the pruned layer's folded copies)
    finally:
        eng.close()


def test_mx2_bucket_space_attention_vs_band_kernel_and_oracle():
    """Round 4: the bucket-space MX attention (attention_mx2.hip: c2p / p2c in delta space, private to each wave; opt-in) on the shapes of the
    test above — against the band kernel of the same engine (same products, other summation order on saturated tiles: accumulation noise)
    and, one row each, against the CPU oracle.  Lengths beyond 512 exercise the log buckets and the saturated ends of the table,
    S = 64 a single query tile, ragged rows the key-length cut.  A developer kernel since round 5 (csrc/dev/, make DEV=1): skipped on the product library."""
    from gliclass.c_amd import _lib
    if not _lib.hip().glc_debug_is_developer_build():
        pytest.skip("attention_mx2 is compiled into developer builds only (make DEV=1)")
    import oracle_c
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["base"]
    eng = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
    w = None
    sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    try:
        eng.set_length_buckets(1)
        for (B, S, Cn, ragged) in ((64, 1024, 8, False), (100, 320, 8, True), (33, 1000, 5, True), (17, 2048, 8, True), (1024, 64, 1, False), (24, 704, 3, True)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=B + S, ragged=ragged)
            eng.set_mx2(False)
            band = eng.forward(ids, mask)
            assert eng.last_mx_attention(), (B, S)
            eng.set_mx2(True)
            got = eng.forward(ids, mask)
            assert eng.last_mx_attention() and np.isfinite(got).all(), (B, S)
            d = float(np.abs(sig(got) - sig(band)).max())
            assert d <= 1e-4, (B, S, Cn, d)                       # measured <= 2e-5: only the saturated tiles are summed in another order
            if S >= 1000:                                        # one whole row against the oracle (the log buckets and both saturated ends)
                if w is None:
                    w = weights.make_weights(cfg, 42)
                b = B // 2
                n = int(mask[b].sum())
                ref = oracle_c.forward(cfg, w, ids[b:b + 1, :n], mask[b:b + 1, :n])
                assert np.abs(sig(got[b:b + 1, :ref.shape[1]]) - sig(ref)).max() <= 5e-4, (B, S)
    finally:
        eng.close()


def test_fp8_range_guard_device_resident_forward_contract():
    """a device-resident forward (glc_engine_forward_device) cannot be repeated behind the caller's back; its logits are
    valid only after glc_engine_sync() == 0 or glc_engine_device_forward_valid() == 1 (include/gliclass_hip.h).  With activation rows
    beyond 448 and tiles inside: the first forward is reported invalid and the engine lowers the rows' exponent; the repeat is valid and
    equals the host-buffer forward; a host forward BETWEEN a device forward and its sync must neither hide nor invent a range error."""
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["small"]
    w = dict(weights.make_weights(cfg, 42))
    # an outlier the LayerNorm gains do not announce (the load-time bound of the proactive guard stays below 448): one channel of every
    # attention-output bias at 600 — the RAW residual sums, which the MX pipeline keeps as operand rows (LayerNorm folded), leave the range
    ch = 77
    for name in list(w):
        if name.endswith("attention.output.dense.bias"):
            bvec = w[name].copy(); bvec[ch] = 600.0; w[name] = bvec
    B, S, Cn = 64, 512, 4
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=9)
    ref_eng = Engine(cfg, w, dtype="f32")
    try:
        ref_eng.set_length_buckets(1)
        assert ref_eng.activation_exponent() == 0          # nothing at load announces the outlier
        want = ref_eng.forward(ids, mask)                   # host-buffer forward: repeats itself, ends at exponent -5 on the MX pipeline
        assert ref_eng.last_mx() and ref_eng.activation_exponent() == -5 and ref_eng.fp8_range_retries() == 1
    finally:
        ref_eng.close()
    eng = Engine(cfg, w, dtype="f32")
    d_ids = d_mask = d_out = None
    try:
        eng.set_length_buckets(1)
        d_ids, d_mask, d_out = eng.dev_alloc(ids.nbytes), eng.dev_alloc(mask.nbytes), eng.dev_alloc(B * Cn * 4)
        eng.h2d(d_ids, ids.astype(np.int64)); eng.h2d(d_mask, mask.astype(np.int64))
        out = np.zeros((B, Cn), np.float32)
        # 1. forward_device + sync: the first one is reported, the engine changes its arithmetic, the repeat is clean
        eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)
        assert eng.L.glc_engine_sync(eng.h) == -1 and b"run the forward again" in eng.L.glc_last_error()
        assert eng.activation_exponent() == -5 and not eng.fp8_range_sticky()
        eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)
        eng.sync()
        eng.d2h(out, d_out)
        assert np.array_equal(out, want)
        # 2. the validity query for callers that wait by other means (here: the D2H copy above has drained the stream)
        eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)
        eng.d2h(out, d_out)
        assert eng.L.glc_engine_device_forward_valid(eng.h) == 1
    finally:
        for p in (d_ids, d_mask, d_out):
            if p: eng.dev_free(p)
        eng.close()
    # 3. a host-buffer forward between a device-resident forward and its sync: the pending verdict survives, no spurious second one
    eng = Engine(cfg, w, dtype="f32")
    d_ids = d_mask = d_out = None
    try:
        eng.set_length_buckets(1)
        d_ids, d_mask, d_out = eng.dev_alloc(ids.nbytes), eng.dev_alloc(mask.nbytes), eng.dev_alloc(B * Cn * 4)
        eng.h2d(d_ids, ids.astype(np.int64)); eng.h2d(d_mask, mask.astype(np.int64))
        eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)          # leaves the range (exponent 0)
        host = eng.forward(ids, mask)                               # settles the pending check first, then runs at exponent -5
        assert np.array_equal(host, want) and eng.activation_exponent() == -5
        assert eng.L.glc_engine_sync(eng.h) == -1                    # the device-resident forward's verdict is still reported ...
        assert eng.L.glc_engine_sync(eng.h) == 0                     # ... once
        eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)
        eng.forward(ids, mask)
        eng.sync()                                                  # a clean device forward + a clean host forward: no invented error
    finally:
        for p in (d_ids, d_mask, d_out):
            if p: eng.dev_free(p)
        eng.close()


def test_product_library_has_no_path_to_timing_only_or_stamped_kernels():
    """the product libgliclass_hip.so (plain `make`) contains no timing-only (wrong-result), stamped or rejected-experiment
    kernel and no switch that reaches one: the attention microbenchmark refuses their variant bits and stamps, the MX GEMM refuses the GY /
    16 x 16 / stamped / ablation requests.  (A developer build — make DEV=1 — has them all: skipped there.)"""
    from gliclass.c_amd import _lib, synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    L = _lib.hip()
    if L.glc_debug_is_developer_build():
        pytest.skip("developer build: the diagnostic kernels are compiled in on purpose")
    cfg = CONFIGS["small"]
    eng = Engine.from_spec(cfg, "synthetic:small:42", dtype="f32")
    try:
        eng.set_length_buckets(1)
        eng.set_group_split(2)
        ids, mask, _ = synth.make_inputs(cfg, 8, 256, 4, seed=5)
        eng.forward(ids, mask)
        assert eng.last_mx_attention()
        cs = (C.c_double * 2)()
        assert L.glc_debug_attn_bench(eng.h, 1, 128, 0, cs) > 0                      # the shipping band kernel
        for bits in (256, 512, 4096, 8192, 16384, 65536, 524288):                   # ablations, PV16, bucket-space kernel, spilled build, 16x16 timing build, two-tiles-per-wave kernel
            assert L.glc_debug_attn_bench(eng.h, 1, 128 | bits, 0, cs) < 0, bits
            assert b"developer builds only" in L.glc_last_error(), bits
        assert L.glc_debug_attn_bench(eng.h, 1, 128, 1, cs) < 0                      # stamps
        out = (C.c_double * 8)()
        for mode in (10, 20, 30):                                                    # GY images, 16 x 16 shapes, (deleted) 128 x 128 wave tile
            assert L.glc_debug_gemm_mx_check(eng.h, 512, 768, 768, 1.0, 0.02, mode, out) != 0, mode
        with pytest.raises(RuntimeError, match="developer builds only"):            # no silent no-op either: the switches themselves refuse
            eng.set_mx2(True)
    finally:
        eng.close()


def test_mxd_two_tiles_per_wave_attention_vs_band_kernel():
    """Round 5, developer kernel (csrc/dev/attention_mxd.hip: two query tiles per wave, one wave per SIMD, software-pipelined; make DEV=1 — skipped on
    the product library): context rows against the band kernel on the SAME MX tiles of a forward, through the attention microbenchmark.  Same products
    except the order in which the c2p band joins the scores and one shared product on saturated tiles: rows of magnitude ~5 must agree to two units of
    the GX output format (2.5e-4; measured 1.2e-4).  Shapes: both saturated ends (S >= 1000), Sp % 256 != 0 (inactive waves), a single tile pair, ragged."""
    from gliclass.c_amd import _lib, synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    L = _lib.hip()
    if not L.glc_debug_is_developer_build():
        pytest.skip("attention_mxd is compiled into developer builds only (make DEV=1)")
    cfg = CONFIGS["base"]
    eng = Engine.from_spec(cfg, "synthetic:base:42", dtype="f32")
    try:
        eng.set_length_buckets(1); eng.set_group_split(2)
        for (B, S, ragged) in ((16, 1024, False), (5, 640, True), (3, 192, True), (2, 64, False), (2, 1536, True)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=3, ragged=ragged)
            L.glc_debug_set_stop(eng.h, 1); eng.forward(ids, mask)
            assert eng.last_mx_attention(), (B, S)
            Sp = (S + 63) // 64 * 64
            rows = B * Sp
            out = {}
            for v in (128, 128 | 524288):
                cs = (C.c_double * 2)()
                assert L.glc_debug_attn_bench(eng.h, 1, v, 0, cs) > 0, L.glc_last_error()
                buf = np.zeros((rows, cfg.hidden), np.float32)
                L.glc_debug_read_workspace(eng.h, 2, rows, buf.ctypes.data_as(C.c_void_p))
                out[v] = buf.reshape(B, Sp, cfg.hidden)
            valid = np.zeros((B, Sp), bool); valid[:, :S] = mask.astype(bool)
            d = float(np.abs(out[128] - out[128 | 524288])[valid].max())
            assert np.isfinite(out[128 | 524288][valid]).all() and d <= 2.5e-4, (B, S, d)
        L.glc_debug_set_stop(eng.h, -1)
    finally:
        eng.close()


def test_resident_position_blocks_bit_identical_to_per_tile_requests():
    """the shipping band kernel keeps a wave's PQ block in registers and refills it in place through untracked inline-asm loads
    (FIXQ, csrc/glc_pfrag.h).  Stale position rows would mostly stay inside the oracle tolerance, so this check does not depend on one: the developer
    library (make DEV=1 devlib -> gliclass/c_amd/variants/, built by __graft_entry__.build()) carries round 4's per-tile request form as variant bit 17;
    context rows of both forms on the SAME MX tiles must be bit-identical — NW = 4 (default) and NW = 8, saturated (S >= 1000), ragged, single-tile and
    Sp % 128 != 0 shapes.  A subprocess: two libraries with the same symbols do not share a process."""
    import subprocess, sys
    dev = os.path.join(ROOT, "gliclass", "c_amd", "variants", "libgliclass_hip_dev.so")
    if not os.path.exists(dev):
        pytest.skip("developer library not built (make -C gliclass/c_amd DEV=1 devlib)")
    env = dict(os.environ, GLC_HIP_SO=dev, GLC_REPS="0", GLC_SHAPES="16x1024,8x512,3x192,2x64,5x640,2x1536")
    # ... and round 6's form of the loop (ready-made offset table, planar position tables, descending c2p ring) against round 5's (variant bit 20): the same
    # products and sums in another instruction stream
    for va, vb in ((128, 128 | 131072), (128 | 2048, 128 | 2048 | 131072), (128, 128 | 1048576), (128 | 2048, 128 | 2048 | 1048576)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "attn_variant_ab.py"), str(va), str(vb)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.count("identical True") == 12, (va, vb, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("gain", [12.0, 40.0, 400.0])
def test_fp8_range_guard_encoder_outlier_channel(gain):
    """VERDICT r3 item 4 / ADVICE r3 (medium): the MX operand images carry e4m3 parts with exponent 0, so an activation beyond 448 saturates
    there and its cross terms silently fall to single-f16 accuracy.  A model with an outlier channel, as trained checkpoints have them: one
    channel of every LayerNorm has gain `gain` (ordinary tokens carry gain * N(0, 1) there), and the embedding rows of every 16th token id
    are dominated by that channel, so those tokens sit at sqrt(H) * gain = 27.7 gain in the normalised rows AND in the raw residual sums of
    every layer: ~3e2 for gain 12 (inside the fp8 range, at its upper end: the MX pipeline must hold its bound), ~1e3 for gain 40 (beyond
    e4m3's 448): every producer counts such elements and the guard's first answer is activation rows with exponent -5 (|x| up to 14336) —
    the forward is repeated ON the MX pipeline and the engine keeps that exponent; ~1e4 for gain 400 (inside f16's 65504; the Q / K / V
    tiles, which keep exponent 0, leave the range too — counted in the guard's second word): the repeat runs on the split-f16 kernels at
    once, and after two such forwards the engine stays there.  The head's projectors ignore the channel (a trained head does not hang
    on an outlier channel either), so the logits stay in the sigmoid's range and the comparison with the oracle means something (bar: the
    reference's own 1e-3, test_onnx.py:30)."""
    import oracle_c
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["small"]
    w = dict(weights.make_weights(cfg, 42))
    ch = 77
    for name in list(w):
        if name.endswith("LayerNorm.weight") and name != "encoder.LayerNorm.weight" and w[name].shape == (cfg.hidden,):      # (not the LayerNorm of the position table)
            g = w[name].copy(); g[ch] = gain; w[name] = g
    emb = w["embeddings.word_embeddings.weight"].copy()
    emb[16:128000:16, ch] = 30.0                      # (word ids only: [CLS], the label / separator tokens stay ordinary)
    w["embeddings.word_embeddings.weight"] = emb
    for pj in ("text_projector", "classes_projector"):
        m = w[pj + ".linear_1.weight"].copy(); m[:, ch] = 0.0; w[pj + ".linear_1.weight"] = m
    B, S, Cn = 40, 512, 4
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=321, ragged=True)
    assert ((ids % 16 == 0) & (ids >= 16) & (ids < 128000)).any()
    sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    eng = Engine(cfg, w, dtype="f32")
    try:
        eng.set_length_buckets(1)
        got = eng.forward(ids, mask)
        assert np.isfinite(got).all()
        if gain < 16:
            assert eng.last_mx() and eng.fp8_range_retries() == 0 and eng.activation_exponent() == 0
        elif gain < 100:
            # round 5: the guard is proactive for checkpoints — gain * sqrt(H) = 905 > 448 is known at load, the engine starts at exponent -5
            # and no forward is repeated (round 4: one repeat)
            assert eng.fp8_range_retries() == 0 and eng.last_mx() and eng.activation_exponent() == -5, "the engine should have started at exponent -5"
            again = eng.forward(ids, mask)
            assert eng.fp8_range_retries() == 0 and eng.last_mx() and not eng.fp8_range_sticky()
            assert np.array_equal(again, got)
        else:
            # round 5: rows and tiles are counted apart — Q / K / V tiles keep exponent 0, so when THEY leave the range the guard
            # goes straight to the split kernels: one repeat, the rows' exponent untouched (a repeat at exponent -5 could not have helped)
            assert eng.fp8_range_retries() == 1 and not eng.last_mx() and eng.activation_exponent() == -5, "tiles beyond the range: straight to the split kernels (exponent -5 from the load-time bound)"
            again = eng.forward(ids, mask)
            assert eng.fp8_range_retries() == 2 and eng.fp8_range_sticky()
            third = eng.forward(ids, mask)
            assert eng.fp8_range_retries() == 2 and not eng.last_mx()         # no MX attempt any more
            assert np.array_equal(again, got) and np.array_equal(third, got)
        worst, span = 0.0, 0.0
        for b in (0, 7, 33):
            n = int(mask[b].sum())
            ref = oracle_c.forward(cfg, w, ids[b:b + 1, :n], mask[b:b + 1, :n])
            worst = max(worst, float(np.abs(sig(got[b:b + 1, :ref.shape[1]]) - sig(ref)).max()))
            span = max(span, float(np.abs(ref).max()))
        print(f"gain {gain:g}: max probability error vs the oracle {worst:.2e}, largest |logit| {span:.2f} (MX pipeline: {eng.last_mx()})")
        assert 0.05 < span < 30, "the logits should sit in the sigmoid's range for the comparison to mean something"
        assert worst <= (5e-4 if gain < 16 else 1e-3), (gain, worst)         # (bar for the outlier models: the reference's own 1e-3)
    finally:
        eng.close()
