"""GPU (-m gpu): the MX cross-term GEMM (gemm256x.hip: a_hi*w_hi in f16 MFMAs + both cross terms in one block-scaled fp8 MFMA, GX rows)
against the split-f16 GEMM (gemm256s.hip, three f16 MFMAs per product, group-split rows) on the same random fp32 operands — every
epilogue path the pipeline uses, through the C-ABI's developer entry glc_debug_gemm_mx_check.  The product error bound of the MX
arithmetic is ~2^-15 relative (cross terms to ~4 bits); measured 1e-5 relative rms.  The whole-forward error of the MX pipeline is
asserted in test_gpu_parity.py (forced pipeline, mini / small) and test_gpu_fullsize.py (c3, three seeds; c4's shard)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = {0: "bias, plain fp32 out", 1: "GELU + LayerNorm fold, row out", 2: "residual (raw rows, LayerNorm on the fly), raw rows + partials out",
         3: "residual, plain fp32 out", 4: "QKV + LayerNorm fold, split-f16 units"}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_mx_gemm_every_epilogue_vs_split_gemm(mode, weights_for):
    from gliclass.c_amd.engine import Engine
    cfg, w = weights_for("tiny")
    eng = Engine(cfg, w, dtype="f16")
    out = (C.c_double * 5)()
    try:
        for (M, N, K, a_amp, w_amp) in ((256, 256, 32, 1.0, 0.05), (512, 768, 768, 2.0, 0.05), (1024, 768, 3072, 1.0, 0.1), (256, 768, 96, 50.0, 1.0)):
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
        for (B, S, Cn, ragged) in ((300, 192, 3, True), (100, 320, 8, True), (33, 1000, 5, True), (70, 448, 1, False), (17, 2048, 8, True), (1024, 64, 1, False)):
            ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=B + S, ragged=ragged)
            got = eng.forward(ids, mask)
            assert eng.last_mx() and np.isfinite(got).all(), (B, S)
            eng.set_mx(False)
            ref = eng.forward(ids, mask)
            eng.set_mx(True)
            assert not np.array_equal(got, ref)
            d = float(np.abs(sig(got) - sig(ref)).max())
            assert d <= 3e-4, (B, S, Cn, d)
    finally:
        eng.close()
