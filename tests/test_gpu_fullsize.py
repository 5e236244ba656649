"""GPU (-m gpu): BASELINE.json's configs at their FULL sizes, through the C-ABI.

  c2  gliclass-small  B=8   S=512    (BASELINE.json configs[1]: the first GPU config, fp32, vs CPU probabilities)
  c3  gliclass-base   B=64  S=1024   (the headline bench shape: the 256x256 staggered GEMM instantiations and the band attention
                                      kernel the bench runs, which smaller batches never dispatch to)
  c4  gliclass-large  B=32  S=1024   (one GPU's shard of the 8-GPU batch of 256)
  c5  qwen-1.5b shape B=16  S=2048   (decoder-style backbone)

The oracle (CPU, fp32) needs seconds per sequence at these lengths, so each case checks a few whole rows against it — rows are
independent end to end, so the oracle runs them as a batch of their own — and the rest of the batch through size-independent
properties: finiteness, and row independence (a row's logits do not depend on its batch mates: the property that makes the batch
shard across GPUs).

Tolerances on per-label probabilities.  The acceptance bar is 1e-3 (north_star; the reference's own check,
/root/reference/ONNX_CONVERTING/test_onnx.py:30).  The DEFAULT mode (GLICLASS_DTYPE=f32: fp32 data, split-f16 MFMA products) is
asserted at 1e-4, ten times inside it.  f16 / bf16 are opt-in throughput modes: their operand rounding alone exceeds the bar on
these synthetic models (docs/LOG_r01-r05.md §2), so they are asserted at their measured envelopes and never used as a parity claim.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BAR = 1e-3
TOL_DEFAULT_MODE = 1e-4      # split-f16 projections (GLICLASS_MX=0, and every forward too small for the 256-tile pipeline)
TOL_MX = 5e-4          # MX arithmetic of large forwards.  Observed worst case 2.4e-4 (soaks of round 5: 7273 + 5051 shapes on the MX pipeline, profiles/r05/soak_*.txt) / 1.8e-4 (profiles/r05/mx_margin_probe.txt: short rows, S = 256-320 —
                        # ~1.0e-4 from the GX-row projections + ~0.8e-4 from the MX-tile attention); asserted at half the bar = 2.1x that (round 4 asserted 3e-4 = 1.36x: a flake
                        # waiting for a seed) against the bar (1e-3, /root/reference/ONNX_CONVERTING/test_onnx.py:30)
TOL_MX_LONG = 3e-4     # ... and on the shapes of BASELINE.json this file runs (c2 ... c5: S >= 512), where the MX arithmetic measures 1e-5 ... 1.2e-4 (profiles/r06/fullsize_values.txt) (rows of 1024+ tokens average the
                        # operand noise out): asserted 3x inside that, so that a 2x precision regression of the headline shape fails here although it would pass TOL_MX
ENVELOPE = {"f16": 1e-2, "bf16": 6e-2}


def sig(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def _tol(dtype, long_rows=True):
    return TOL_MX_LONG if dtype == "f32" else ENVELOPE[dtype]


def _check_rows_vs_oracle(cfg, w, ids, mask, got, rows, tol):
    import oracle_c
    ref = oracle_c.forward(cfg, w, ids[rows], mask[rows])
    err = float(np.abs(sig(got[rows]) - sig(ref)).max())
    assert err <= tol, (rows, err)
    return err


def test_c1_small_b1_s128(c_generated_weights):
    """BASELINE.json configs[0] at its exact shape: gliclass-small, batch 1, seq 128, 4 labels (the reference's own CPU-runnable case) — the
    one row vs the oracle in the default mode and in the 16-bit modes at their envelopes; a small forward (split-f16 arithmetic, 128-tile GEMMs)."""
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["small"]
    spec = "synthetic:small:42"
    w = c_generated_weights(spec, cfg)
    ids, mask, _ = synth.make_inputs(cfg, 1, 128, 4, seed=1234)
    for dtype in ("f32", "f16"):
        eng = Engine.from_spec(cfg, spec, dtype=dtype)
        try:
            got = eng.forward(ids, mask)
            assert got.shape == (1, 4) and np.isfinite(got).all()
            if dtype == "f32":
                assert not eng.last_mx()
            err = _check_rows_vs_oracle(cfg, w, ids, mask, got, [0], TOL_DEFAULT_MODE if dtype == "f32" else _tol(dtype))
            print(f"c1 {dtype}: max |prob - oracle| = {err:.2e} (bar {BAR})")
        finally:
            eng.close()


def test_c2_small_b8_s512(c_generated_weights):
    """BASELINE.json configs[1] at its exact shape: gliclass-small, batch 8, seq 512, fp32 (the default mode) — all 8 rows vs the oracle."""
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["small"]
    spec = "synthetic:small:42"
    w = c_generated_weights(spec, cfg)
    B, S, Cn = 8, 512, 8
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234)
    eng = Engine.from_spec(cfg, spec, dtype="f32")
    try:
        eng.set_length_buckets(1)
        got = eng.forward(ids, mask)
        assert got.shape == (B, Cn) and np.isfinite(got).all()
        err = _check_rows_vs_oracle(cfg, w, ids, mask, got, list(range(B)), TOL_DEFAULT_MODE)
        print(f"c2 f32: max |prob - oracle| on all 8 rows = {err:.2e} (bar {BAR})")
    finally:
        eng.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_c3_base_b64_s1024(dtype, c_generated_weights):
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["base"]
    spec = "synthetic:base:42"
    w = c_generated_weights(spec, cfg)
    B, S, Cn = 64, 1024, 8
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234)          # the bench's own batch (rank 0)
    eng = Engine.from_spec(cfg, spec, dtype=dtype)
    try:
        eng.set_length_buckets(1)                                       # one forward of M = 65 536 rows, as the bench runs it
        got = eng.forward(ids, mask)
        assert got.shape == (B, Cn) and np.isfinite(got).all()
        err = _check_rows_vs_oracle(cfg, w, ids, mask, got, [0, 37, 63], _tol(dtype, True))
        print(f"c3 {dtype}: max |prob - oracle| on 3 rows = {err:.2e} (bar {BAR})")
        if dtype == "f32":
            assert err <= BAR and eng.last_mx()                     # the default arithmetic of this shape: MX cross-term projections
            eng.set_mx(False)                                       # ... and the split-f16 projections (GLICLASS_MX=0), ten times inside the bar
            exact = eng.forward(ids, mask)
            assert not eng.last_mx()
            e2 = _check_rows_vs_oracle(cfg, w, ids, mask, exact, [0, 37, 63], TOL_DEFAULT_MODE)
            assert np.abs(sig(got) - sig(exact)).max() <= TOL_MX_LONG    # all 512 probabilities, mode against mode
            print(f"c3 f32, GLICLASS_MX=0: {e2:.2e}; MX vs split over all 512 probabilities {np.abs(sig(got) - sig(exact)).max():.2e}")
            eng.set_mx(True)
            # the WHOLE batch against the oracle once (all 64 rows = 512 probabilities; ~1 minute of host cores)
            e_all = _check_rows_vs_oracle(cfg, w, ids, mask, got, list(range(B)), TOL_MX_LONG)
            print(f"c3 f32: all {B} rows vs the oracle: max |prob - oracle| = {e_all:.2e} (asserted {TOL_MX_LONG}, bar {BAR})")
        # row independence across the batch: the upper half alone (M = 32 768: other tile counts, same rows)
        half = eng.forward(ids[32:], mask[32:])
        assert np.abs(sig(half) - sig(got[32:])).max() <= (1e-5 if dtype == "f32" else 5e-3)
        # and permutation of the rows permutes the logits exactly (same shapes => same kernels, bit for bit)
        perm = np.random.RandomState(3).permutation(B)
        assert np.array_equal(eng.forward(ids[perm], mask[perm]), got[perm])
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [43, 44])
def test_c3_default_mode_other_weight_seeds(seed, c_generated_weights):
    """The default mode's error is a property of the weights as much as of the arithmetic (round 3 measured 3 seeds x 512
    probabilities for every candidate operand format, scripts/precision_budget.py): the headline shape with two more synthetic models —
    all 512 probabilities of the MX default against the split-f16 projections, one whole row of each against the oracle."""
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["base"]
    spec = f"synthetic:base:{seed}"
    w = c_generated_weights(spec, cfg)
    B, S, Cn = 64, 1024, 8
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234)
    eng = Engine.from_spec(cfg, spec, dtype="f32")
    try:
        eng.set_length_buckets(1)
        got = eng.forward(ids, mask)
        assert eng.last_mx() and np.isfinite(got).all()
        eng.set_mx(False)
        exact = eng.forward(ids, mask)
        d = float(np.abs(sig(got) - sig(exact)).max())
        err = _check_rows_vs_oracle(cfg, w, ids, mask, got, [11], TOL_MX_LONG)
        e2 = _check_rows_vs_oracle(cfg, w, ids, mask, exact, [11], TOL_DEFAULT_MODE)
        print(f"c3 f32 seed {seed}: MX vs split over 512 probabilities {d:.2e}; row 11 vs oracle: MX {err:.2e}, split {e2:.2e}")
        assert d <= TOL_MX_LONG
    finally:
        eng.close()


@pytest.mark.parametrize("scorer", ["weighted-dot", "mlp"])
def test_c3_shape_with_other_scorers(scorer, c_generated_weights):
    """The headline shape with the head's other scorers (row a12): the whole default pipeline — MX projections and attention, the pruned
    last layer, the compacted head rows — feeding `weighted-dot` / `mlp`; two rows against the oracle, all 512 probabilities MX against split."""
    import dataclasses
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS, SCORER_NAMES
    from gliclass.c_amd.engine import Engine
    cfg = dataclasses.replace(CONFIGS["base"], scorer=SCORER_NAMES[scorer])
    spec = f"synthetic:base:42:{scorer}"
    w = c_generated_weights(spec, cfg)
    B, S, Cn = 64, 1024, 8
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234)
    eng = Engine.from_spec(cfg, spec, dtype="f32")
    try:
        eng.set_length_buckets(1)
        got = eng.forward(ids, mask)
        assert got.shape == (B, Cn) and np.isfinite(got).all() and eng.last_mx()
        err = _check_rows_vs_oracle(cfg, w, ids, mask, got, [5, 58], TOL_MX_LONG)
        eng.set_mx(False)
        exact = eng.forward(ids, mask)
        d = float(np.abs(sig(got) - sig(exact)).max())
        print(f"c3 {scorer}: rows 5, 58 vs oracle {err:.2e}; MX vs split over 512 probabilities {d:.2e}; logit range [{got.min():.2f}, {got.max():.2f}]")
        assert d <= TOL_MX_LONG
    finally:
        eng.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_c4_large_shard_b32_s1024(dtype, c_generated_weights):
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["large"]
    spec = "synthetic:large:42"
    w = c_generated_weights(spec, cfg)
    B, S, Cn = 32, 1024, 8
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234)
    eng = Engine.from_spec(cfg, spec, dtype=dtype)
    try:
        eng.set_length_buckets(1)
        got = eng.forward(ids, mask)
        assert got.shape == (B, Cn) and np.isfinite(got).all()
        err = _check_rows_vs_oracle(cfg, w, ids, mask, got, [5], _tol(dtype))
        print(f"c4 shard {dtype}: max |prob - oracle| on 1 row = {err:.2e} (bar {BAR})")
        if dtype == "f32":                                               # the split-f16 arithmetic (GLICLASS_MX=0) of the same shape: ten times inside the bar
            assert eng.last_mx()
            eng.set_mx(False)
            exact = eng.forward(ids, mask)
            assert not eng.last_mx()
            e2 = _check_rows_vs_oracle(cfg, w, ids, mask, exact, [5], TOL_DEFAULT_MODE)
            print(f"c4 shard f32, GLICLASS_MX=0: {e2:.2e}; MX vs split over all {B * Cn} probabilities {np.abs(sig(got) - sig(exact)).max():.2e}")
            assert np.abs(sig(got) - sig(exact)).max() <= TOL_MX_LONG
            eng.set_mx(True)
        quarter = eng.forward(ids[8:16], mask[8:16])
        assert np.abs(sig(quarter) - sig(got[8:16])).max() <= (1e-5 if dtype == "f32" else 5e-3)
    finally:
        eng.close()


def test_c4_global_batch_256_as_eight_shards_on_one_gpu(c_generated_weights):
    """BASELINE config c4 at its GLOBAL size on the real engine: gliclass-large, 256 rows of 1024 tokens — the batch the
    reference's loop walks in chunks (/root/reference/main.c:141-150, src/parallel_processor.c:28-45) and this build partitions over 8 GPUs
    (bench.py shard_rows).  One GPU plays all eight ranks: shard g = rows shard_rows(256, 8, g), each forwarded on its
    own, concatenated in rank order — (i) equal to the one-piece forward of all 256 rows (every row is independent end to end: the
    partition must not be visible), (ii) rows 0 / 100 / 255 against the CPU oracle, (iii) shard independence: a shard forwarded behind a
    different shard, and with its rows reversed, gives the same probabilities row by row."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import shard_rows
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["large"]
    spec = "synthetic:large:42"
    w = c_generated_weights(spec, cfg)
    B, S, Cn, G = 256, 1024, 8, 8
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=4321)
    eng = Engine.from_spec(cfg, spec, dtype="f32")
    try:
        eng.set_length_buckets(1)
        parts = []
        for g in range(G):
            lo, hi = shard_rows(B, G, g)
            assert hi - lo == B // G
            parts.append(eng.forward(ids[lo:hi], mask[lo:hi]))
            assert eng.last_mx()
        stitched = np.concatenate(parts, axis=0)
        assert stitched.shape == (B, Cn) and np.isfinite(stitched).all()
        whole = eng.forward(ids, mask)                                     # the same 256 rows in one piece
        assert eng.last_mx()
        d = float(np.abs(sig(whole) - sig(stitched)).max())
        print(f"c4 global batch: 8 stitched shards vs one piece, max |prob diff| = {d:.2e}")
        assert d <= 1e-5
        err = _check_rows_vs_oracle(cfg, w, ids, mask, stitched, [0, 100, 255], TOL_MX_LONG)
        print(f"c4 global batch: rows 0 / 100 / 255 vs the oracle: {err:.2e} (asserted {TOL_MX_LONG}, bar {BAR})")
        lo, hi = shard_rows(B, G, 5)
        again = eng.forward(ids[lo:hi], mask[lo:hi])                       # behind the one-piece forward instead of behind shard 4
        assert np.abs(sig(again) - sig(parts[5])).max() <= 1e-6
        rev = eng.forward(ids[lo:hi][::-1].copy(), mask[lo:hi][::-1].copy())
        assert np.abs(sig(rev[::-1]) - sig(parts[5])).max() <= 1e-5
    finally:
        eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_c5_decoder_b16_s2048(dtype, c_generated_weights):
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["qwen-1.5b"]
    spec = "synthetic:qwen-1.5b:42"
    B, S, Cn = 16, 2048, 8
    ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234)
    eng = Engine.from_spec(cfg, spec, dtype=dtype)
    try:
        eng.set_length_buckets(1)
        got = eng.forward(ids, mask)
        assert got.shape == (B, Cn) and np.isfinite(got).all()
        was_mx = eng.last_mx()
        pair = eng.forward(ids[6:8], mask[6:8])                          # row independence: two rows on their own
        # (f32: the two-row forward is small enough to take the plain-fp32 row format and the 128-tile kernels, the batch the
        #  group-split format and the 256-tile kernels: same arithmetic, other summation order — both sit ~1e-5 from the oracle)
        # (round 3: the batch runs the MX cross-term projections, the two-row forward the split-f16 ones)
        assert np.abs(sig(pair) - sig(got[6:8])).max() <= (TOL_MX_LONG if dtype == "f32" else 2e-2)
        if dtype == "f32":                                               # one whole row of 2048 tokens against the oracle
            w = c_generated_weights(spec, cfg)
            err = _check_rows_vs_oracle(cfg, w, ids, mask, got, [3], TOL_MX_LONG)
            print(f"c5 f32: max |prob - oracle| on 1 row = {err:.2e} (bar {BAR})")
            assert was_mx
            eng.set_mx(False)                                            # the split-f16 projections (GLICLASS_MX=0): ten times inside the bar
            exact = eng.forward(ids, mask)
            assert not eng.last_mx()
            e2 = _check_rows_vs_oracle(cfg, w, ids, mask, exact, [3], TOL_DEFAULT_MODE)
            print(f"c5 f32, GLICLASS_MX=0: {e2:.2e}; MX vs split over all {B * Cn} probabilities {np.abs(sig(got) - sig(exact)).max():.2e}")
            assert np.abs(sig(got) - sig(exact)).max() <= TOL_MX_LONG
    finally:
        eng.close()
