"""GPU (-m gpu): the decoder-style backbone (BASELINE.json configs[4]: RoPE + grouped-query causal attention + SwiGLU +
RMSNorm) against the C oracle, which tests/test_oracle.py pins on transformers' Qwen2Model through
tests/golden/dec_*.npz.  The fixtures themselves also run through test_gpu_parity.py::test_golden_fixtures.
Tolerances: see test_gpu_parity.py (fp32 is the parity-grade mode; 16-bit modes are asserted at their measured envelope)."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_PROB = {"f32": 1e-4, "f16": 1e-2, "bf16": 6e-2}


def sig(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


_SWEEP = ((5, 33, 3, [3, 0, 1, 2, 3], 11), (1, 129, 1, None, 12), (3, 64, 2, None, 13), (2, 515, 4, [4, 1], 14))


@pytest.fixture(scope="module")
def sweep_refs(weights_for):
    """Oracle logits of the sweep shapes, computed once for the three operand types."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for("dec-mini")
    out = []
    for (B, S, Cn, lpr, seed) in _SWEEP:
        ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=seed, ragged=True, labels_per_row=lpr)
        out.append((ids, mask, oracle_c.forward(cfg, w, ids, mask)))
    return out


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_decoder_live_oracle_sweep(dtype, weights_for, sweep_refs):
    """Shapes outside the fixtures: S not a multiple of 64, B = 1, rows without labels, S = 1."""
    from gliclass.c_amd.engine import Engine
    cfg, w = weights_for("dec-mini")
    eng = Engine(cfg, w, dtype=dtype)
    try:
        for (ids, mask, ref) in sweep_refs:
            B, S = ids.shape
            got = eng.forward(ids, mask)
            assert got.shape == ref.shape and np.isfinite(got).all()
            assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB[dtype], (B, S)
        ids = np.array([[cfg.cls_id]], np.int64)
        assert eng.forward(ids, np.ones_like(ids), c_alloc=0).shape == (1, 0)
    finally:
        eng.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_decoder_head_dim_64_and_long_sequence(dtype, weights_for):
    """head_dim 64 (the D = 64 instantiation of the layout pass and of the MFMA kernel) with 4 query / 2 kv heads, and a
    sequence longer than any fixture (S = 2500: RoPE table, 79 key tiles, diagonal tile in the last partial 64-block)."""
    import oracle_c
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.engine import Engine
    base, _ = weights_for("dec-tiny")
    cfg = dataclasses.replace(base, head_dim=64, heads=4, kv_heads=2, layers=2)
    w = weights.make_weights(cfg, 11)
    eng = Engine(cfg, w, dtype=dtype)
    try:
        for (B, S, seed) in ((3, 150, 1), (1, 2500, 2)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, 3, seed=seed, ragged=True)
            ref = oracle_c.forward(cfg, w, ids, mask)
            got = eng.forward(ids, mask)
            assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB[dtype], (B, S)
            if dtype == "f32":      # the D = 64 instantiations of the MX pipeline's layout pass and LDS-ring attention (decoder_mx.hip; no RoPE epilogue at head_dim 64)
                eng.set_group_split(2)
                mx = eng.forward(ids, mask)
                eng.set_group_split(1)
                assert eng.last_mx() and eng.last_mx_attention(), "the MX pipeline did not run"
                assert np.abs(sig(mx) - sig(ref)).max() <= 5e-4, (B, S)
    finally:
        eng.close()


@pytest.mark.parametrize("variant", ["bidirectional", "first-pooling", "mha"])
def test_decoder_config_switches(variant, weights_for):
    """causal = 0 (the LLM2Vec-style bidirectional wrapping, unpinned upstream), pooling = 'first', and kv_heads = heads
    are config switches of the same kernels; checked against the oracle in fp32."""
    import oracle_c
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.config import POOL_FIRST
    from gliclass.c_amd.engine import Engine
    base, _ = weights_for("dec-tiny")
    cfg = {"bidirectional": dataclasses.replace(base, causal=0),
           "first-pooling": dataclasses.replace(base, pooling=POOL_FIRST, causal=0),
           "mha": dataclasses.replace(base, kv_heads=base.heads)}[variant]
    w = weights.make_weights(cfg, 7)
    eng = Engine(cfg, w, dtype="f32")
    try:
        for (B, S, seed) in ((3, 100, 1), (2, 300, 2)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, 3, seed=seed, ragged=True)
            ref, hid = oracle_c.forward(cfg, w, ids, mask, want_hidden=True)
            eng.keep_hidden(True)
            got = eng.forward(ids, mask)
            eng.keep_hidden(False)
            assert np.abs(sig(got) - sig(ref)).max() <= 1e-4
            m = mask.astype(bool)
            for which in range(cfg.layers + 1):
                assert np.abs(eng.hidden(which, B, S)[m] - hid[which][m]).max() <= 3e-4, which
            # the same switches on the forced MX pipeline: non-causal walks of the LDS-ring attention, one query head per kv head ("mha":
            # two kv heads, so the QKV projection also takes the RoPE epilogue), first-token pooling
            eng.set_group_split(2)
            mx = eng.forward(ids, mask)
            eng.set_group_split(1)
            assert eng.last_mx() and eng.last_mx_attention(), "the MX pipeline did not run"
            assert np.abs(sig(mx) - sig(ref)).max() <= 5e-4, (variant, B, S)
    finally:
        eng.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("causal", [1, 0])
def test_decoder_mfma_attention_equals_simple_attention(dtype, causal, weights_for):
    """The flash-style MFMA kernel (fragment-major operands, RoPE applied by the layout pass) and the straightforward kernel
    (row-major operands, RoPE in place) see the same rounded Q/K/V, so layer outputs agree to accumulation-order noise —
    including the diagonal (causal) tiles, ragged key lengths and S that is not a multiple of 64."""
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.engine import Engine
    base, _ = weights_for("dec-mini")
    cfg = dataclasses.replace(base, causal=causal)
    w = weights.make_weights(cfg, 42)
    eng = Engine(cfg, w, dtype=dtype)
    try:
        for (B, S, seed) in ((3, 77, 1), (2, 640, 2), (1, 1100, 3)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, 2, seed=seed, ragged=True)
            outs = []
            for impl in (1, 2):
                eng.set_attention_impl(impl)
                eng.keep_hidden(True)
                eng.forward(ids, mask)
                outs.append(eng.hidden(1, B, S))
            eng.set_attention_impl(0)
            eng.keep_hidden(False)
            m = mask.astype(bool)
            tol = 2e-2 if dtype == "f16" else 1.5e-1
            assert np.abs(outs[0][m] - outs[1][m]).max() <= tol, (B, S)
    finally:
        eng.close()


def test_decoder_rows_are_independent(weights_for):
    """A row's logits do not depend on its batch neighbours or on the padding it is batched with."""
    from gliclass.c_amd import synth
    from gliclass.c_amd.engine import Engine
    cfg, w = weights_for("dec-tiny")
    eng = Engine(cfg, w, dtype="f32")
    try:
        ids, mask, _ = synth.make_inputs(cfg, 4, 150, 3, seed=5, ragged=True)
        full = eng.forward(ids, mask)
        for b in range(4):
            n = int(mask[b].sum())
            solo = eng.forward(ids[b:b + 1, :n], mask[b:b + 1, :n])
            k = solo.shape[1]
            assert np.abs(solo[0] - full[b, :k]).max() <= 2e-5
    finally:
        eng.close()


def test_decoder_through_model_h(weights_for):
    """The drop-in surface (include/model.h) serves a decoder backbone unchanged: create_ort_session on a synthetic
    decoder model -> prepare_input_tensors -> run_inference (the reference call sequence, /root/reference/main.c:83-99,141-150)."""
    import os
    import oracle_c
    from gliclass.c_amd import _lib, synth
    cfg, w = weights_for("dec-tiny")
    m = _lib.model()
    os.environ["GLICLASS_DTYPE"] = "f32"
    try:
        m.initialize_ort_api()
        sess = m.create_ort_session(m.initialize_ort_environment(), b"synthetic:dec-tiny:42", 8)
        assert sess
        ids, mask, _ = synth.make_inputs(cfg, 3, 90, 2, seed=9, ragged=True)
        i32, m32 = ids.astype(np.int32), mask.astype(np.int32)
        rows_i = (C.POINTER(C.c_int) * 3)(*[i32[b].ctypes.data_as(C.POINTER(C.c_int)) for b in range(3)])
        rows_m = (C.POINTER(C.c_int) * 3)(*[m32[b].ctypes.data_as(C.POINTER(C.c_int)) for b in range(3)])
        tok = _lib.TokenizedInputs(rows_i, rows_i, rows_m, 3, 90)
        a, b = C.POINTER(_lib.OrtValue)(), C.POINTER(_lib.OrtValue)()
        assert m.prepare_input_tensors(C.byref(tok), C.byref(a), C.byref(b)) == 0
        out = m.run_inference(sess, a, b)
        ref = oracle_c.forward(cfg, w, ids, mask)
        assert out and list(out.contents.dims[:2]) == [3, ref.shape[1]]
        got = np.ctypeslib.as_array(C.cast(out.contents.data, C.POINTER(C.c_float)), shape=ref.shape).copy()
        assert np.abs(sig(got) - sig(ref)).max() <= 1e-4
    finally:
        del os.environ["GLICLASS_DTYPE"]


def test_c5_full_size_decoder(weights_for, c_generated_weights):
    """BASELINE.json configs[4] at its real size (Qwen2-1.5B shape: 28 layers, H 1536, 12 query / 2 kv heads of 128, SwiGLU 8960,
    vocab 151 648; S = 2048).  The oracle cannot run S = 2048 in seconds, so at full length the test uses size-independent
    properties — causality (a hidden state never depends on later tokens: bit-identical prefix), row independence, finiteness
    — and pins the 1.5 B-parameter weight path on the oracle at a short sequence."""
    import oracle_c
    from gliclass.c_amd import _lib, synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["qwen-1.5b"]
    spec = "synthetic:qwen-1.5b:42"
    eng = Engine.from_spec(cfg, spec, dtype="bf16")
    try:
        # (1) full length: causality + row independence
        B, S = 2, 2048
        ids, mask, _ = synth.make_inputs(cfg, B, S, 8, seed=77, ragged=False)
        eng.keep_hidden(True)
        base = eng.forward(ids, mask)
        h_base = eng.hidden(cfg.layers, B, S)
        assert np.isfinite(base).all() and np.isfinite(h_base).all()
        cut = 1504      # a multiple of 32: queries that share a 32-row wave tile also share its wave-uniform rescale decisions,
        ids2 = ids.copy()   # so only whole earlier tiles are bit-identical (the partial tile differs by rounding, not by content)
        ids2[0, cut:] = (ids2[0, cut:] * 7 + 13) % 1000 + 3               # different tokens after the cut, row 0 only
        eng.forward(ids2, mask)
        h_mod = eng.hidden(cfg.layers, B, S)
        eng.keep_hidden(False)
        assert np.array_equal(h_mod[0, :cut], h_base[0, :cut])             # prefix never sees the suffix
        assert np.array_equal(h_mod[1], h_base[1])                         # the other row is untouched
        assert not np.array_equal(h_mod[0, cut:], h_base[0, cut:])
        solo = eng.forward(ids[1:2], mask[1:2])
        assert np.abs(sig(solo[0]) - sig(base[1])).max() <= 2e-2           # batch neighbour changes tile scheduling only (bf16)
        # (2) the 1.5 B weight path against the oracle at a short sequence
        w = c_generated_weights(spec, cfg)
        if True:
            ids_s, mask_s, _ = synth.make_inputs(cfg, 1, 48, 3, seed=5, ragged=False)
            ref = oracle_c.forward(cfg, w, ids_s, mask_s)
            got = eng.forward(ids_s, mask_s)
            assert np.abs(sig(got) - sig(ref)).max() <= 6e-2               # bf16 envelope (TOL_PROB)
    finally:
        eng.close()


@pytest.mark.parametrize("cname", ["dec-tiny", "dec-mini"])
def test_decoder_group_split_pipeline_vs_oracle_and_plain_fp32(cname, weights_for):
    """fp32 mode of the decoder backbone, the two activation formats (see test_gpu_parity.py::test_group_split_pipeline_...): group-split
    RMSNorm / context / SwiGLU rows + 256-tile GEMMs (plain fp32 residual stream, QKV rows and [gate | up] rows) against the plain
    fp32 rows + 128-tile split GEMMs and against the oracle."""
    import oracle_c
    from gliclass.c_amd import synth
    from gliclass.c_amd.engine import Engine
    cfg, w = weights_for(cname)
    eng = Engine(cfg, w, dtype="f32")
    try:
        for (B, S, Cn, lpr, seed) in ((3, 200, 3, [3, 0, 2], 71), (2, 700, 4, None, 72), (5, 64, 2, None, 73)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=seed, ragged=True, labels_per_row=lpr)
            ref = oracle_c.forward(cfg, w, ids, mask)
            eng.set_group_split(0)
            plain = eng.forward(ids, mask)
            assert not eng.last_group_split()
            eng.set_group_split(2)
            eng.set_mx(False)                 # the split-f16 projections; the MX pipeline is checked at the end of the loop body
            gs = eng.forward(ids, mask)
            assert eng.last_group_split() and not eng.last_mx()
            assert np.isfinite(gs).all()
            assert np.abs(sig(gs) - sig(ref)).max() <= TOL_PROB["f32"], (B, S)
            assert np.abs(sig(gs) - sig(plain)).max() <= 1e-4, (B, S)
            # RMSNorm folded into the GEMMs (default: raw group-split residual stream + (0, rstd) per row, gain folded into the
            # weights, rstd applied in the QKV / SwiGLU epilogues) against RMSNorm as kernels of its own
            eng.set_ln_fused(False)
            unf = eng.forward(ids, mask)
            eng.set_ln_fused(True)
            assert eng.last_group_split()
            assert np.abs(sig(unf) - sig(ref)).max() <= TOL_PROB["f32"], (B, S)
            assert np.abs(sig(gs) - sig(unf)).max() <= 1e-4, (B, S)
            if cfg.hidden % 256 == 0:
                assert not np.array_equal(gs, unf), "the RMSNorm switch changed nothing: is the folded path running?"
                # MX cross-term projections (round 3; the default of large decoder forwards too): GX rows + gemm256x incl. its SwiGLU epilogue
                eng.set_mx(True)
                mx = eng.forward(ids, mask)
                assert eng.last_mx(), "the MX pipeline did not run"
                assert np.isfinite(mx).all() and not np.array_equal(mx, gs)
                assert np.abs(sig(mx) - sig(ref)).max() <= 5e-4, (B, S)
                assert np.abs(sig(mx) - sig(gs)).max() <= 5e-4, (B, S)
                # dec-mini (head_dim 128, even head counts): RoPE + MX tiles are the QKV projection's epilogue (gemm256x EPI_QKVR, weight
                # rows of every Q / K head reordered at load).  With the attention back on split units the same reordered weights go through
                # the plain-row epilogue (columns put back in place) and the separate RoPE / layout pass: both orders, one answer.
                assert eng.last_mx_attention()
                eng.set_mx_attention(False)
                mxs = eng.forward(ids, mask)
                eng.set_mx_attention(True)
                assert eng.last_mx() and not eng.last_mx_attention()
                assert np.abs(sig(mxs) - sig(ref)).max() <= 5e-4, (B, S)
                assert np.abs(sig(mxs) - sig(mx)).max() <= 5e-4, (B, S)
    finally:
        eng.close()


def test_decoder_massive_activation_channel_falls_back_to_unfused_norms(weights_for):
    """with RMSNorm folded into the GEMMs the RAW residual stream is a split-f16 GEMM operand.  A pre-norm decoder
    whose residual stream carries a massive-activation channel (|x| > 65504: beyond f16, although every normalised row is tiny)
    makes that forward non-finite; the engine must then repeat it once with the norms as kernels of their own (plain fp32 residual
    stream, only normalised rows split) and return what the fp32 reference returns."""
    import oracle_c
    from gliclass.c_amd import synth
    from gliclass.c_amd.engine import Engine
    cfg, w0 = weights_for("dec-mini")
    w = dict(w0)
    emb = w0["embed_tokens.weight"].copy()
    emb[:, 5] *= 2.0e5                         # one channel of the embedding rows ~ 1e5: the residual stream leaves the f16 range
    w["embed_tokens.weight"] = emb
    ids, mask, _ = synth.make_inputs(cfg, 3, 200, 3, seed=91, ragged=True)
    ref = oracle_c.forward(cfg, w, ids, mask)
    assert np.isfinite(ref).all()
    eng = Engine(cfg, w, dtype="f32")
    try:
        eng.set_group_split(2)
        got = eng.forward(ids, mask)
        assert eng.last_group_split()
        assert np.isfinite(got).all()
        if cfg.hidden % 256 == 0:
            assert eng.range_retries() == 1, "the folded forward should have overflowed and been repeated"
        assert np.abs(sig(got) - sig(ref)).max() <= 1e-3
    finally:
        eng.close()


@pytest.mark.parametrize("amp", [150.0, 2.0e3, 2.0e4])
def test_decoder_outlier_tokens_fp8_range_guard(amp, weights_for):
    """VERDICT r3 item 4 / ADVICE r3 (medium), decoder backbone: with RMSNorm folded, the RAW residual stream is a GX operand of the MX
    projections, and a pre-norm decoder carries massive activations at a few tokens (10^3 .. 10^4 in one channel, inside f16's range,
    beyond e4m3's 448).  Here the embedding rows of every 16th token id carry +-amp in one channel — the residual stream keeps it through
    every layer.  amp = 150 stays inside the fp8 range: the MX pipeline must hold its bound; beyond 448 the producers count the element and
    the forward is repeated — first on the MX pipeline with activation rows of exponent -5 (2e3: that holds it), then, if the range is left
    again (2e4 > 14336), on the split-f16 kernels (not the unfused-norm retry: nothing overflowed) — and must match the oracle."""
    import oracle_c
    from gliclass.c_amd import synth
    from gliclass.c_amd.engine import Engine
    cfg, w0 = weights_for("dec-mini")
    w = dict(w0)
    emb = w0["embed_tokens.weight"].copy()
    emb[16::16, 5] = amp
    emb[24::32, 5] = -amp
    w["embed_tokens.weight"] = emb
    ids, mask, _ = synth.make_inputs(cfg, 3, 200, 3, seed=91, ragged=True)
    assert ((ids % 16) == 0).any()
    ref = oracle_c.forward(cfg, w, ids, mask)
    assert np.isfinite(ref).all()
    eng = Engine(cfg, w, dtype="f32")
    try:
        eng.set_group_split(2)
        got = eng.forward(ids, mask)
        assert eng.last_group_split() and np.isfinite(got).all()
        assert eng.range_retries() == 0
        if amp < 400:
            assert eng.last_mx() and eng.fp8_range_retries() == 0 and eng.activation_exponent() == 0
        elif amp < 1.4e4:
            assert eng.fp8_range_retries() == 1 and eng.last_mx() and eng.activation_exponent() == -5
        else:
            assert eng.fp8_range_retries() == 2 and not eng.last_mx()
        err = float(np.abs(sig(got) - sig(ref)).max())
        print(f"amp {amp:g}: max probability error vs the oracle {err:.2e} (MX pipeline: {eng.last_mx()})")
        assert err <= (5e-4 if amp < 400 else 1e-3), (amp, err)
    finally:
        eng.close()


@pytest.mark.parametrize("amp", [3000.0, 30000.0])
def test_fp8_range_guard_device_resident_path(amp, weights_for):
    """The device-resident forward (glc_engine_forward_device, what bench.py times) cannot be repeated behind the caller's back: when its
    operands left the fp8 range, glc_engine_sync must FAIL with a message — never hand out such logits — and the engine's next forward must
    be good: on the MX pipeline with activation rows of exponent -5 when that holds the outliers (3000), on the split-f16 kernels after a
    second failed sync when it does not (30000 > 14336)."""
    import oracle_c
    from gliclass.c_amd import synth
    from gliclass.c_amd.engine import Engine
    cfg, w0 = weights_for("dec-mini")
    w = dict(w0)
    emb = w0["embed_tokens.weight"].copy()
    emb[16::16, 5] = amp
    w["embed_tokens.weight"] = emb
    B, S, Cn = 3, 200, 3
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=91, ragged=True)
    ref = oracle_c.forward(cfg, w, ids, mask)
    eng = Engine(cfg, w, dtype="f32")
    d_ids = d_mask = d_out = None
    try:
        eng.set_group_split(2)
        d_ids, d_mask, d_out = eng.dev_alloc(ids.nbytes), eng.dev_alloc(mask.nbytes), eng.dev_alloc(B * Cn * 4)
        eng.h2d(d_ids, ids); eng.h2d(d_mask, mask)
        eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)
        assert eng.last_mx()
        with pytest.raises(RuntimeError, match="fp8 range"):
            eng.sync()
        assert eng.activation_exponent() == -5 and not eng.fp8_range_sticky()
        eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)       # exponent -5: holds 3000, not 30000
        assert eng.last_mx()
        if amp > 14336:
            with pytest.raises(RuntimeError, match="fp8 range"):
                eng.sync()
            assert eng.fp8_range_sticky()
            eng.forward_device(d_ids, d_mask, B, S, Cn, d_out)   # the engine has left the MX pipeline: this one is good
            eng.sync()
            assert not eng.last_mx()
        else:
            eng.sync()
        got = np.zeros((B, Cn), np.float32)
        eng.d2h(got, d_out)
        assert np.abs(sig(got) - sig(ref)).max() <= 1e-3
    finally:
        for p in (d_ids, d_mask, d_out):
            if p:
                eng.dev_free(p)
        eng.close()
