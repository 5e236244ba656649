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
