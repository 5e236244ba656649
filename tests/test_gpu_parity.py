"""GPU (-m gpu): parity of the HIP path against the oracle and the committed golden fixtures, all
through the C-ABI (include/gliclass_hip.h and include/model.h).

Tolerances (per-label probabilities, the quantity north_star bounds):
  f32  operands: 1e-3 is the bar (BASELINE.json); measured ~1e-6, asserted at 1e-4.
  f16 / bf16 operands (the MFMA throughput modes): every GEMM/attention operand is rounded to 11 / 8
  significant bits.  On the deliberately sensitive synthetic models this alone moves logits by ~1e-2 (f16)
  — a CPU emulation of the same rounding points reproduces that magnitude with NO single dominant source
  (scripts/emulate_rounding.py, profiles/r01_f16_rounding_ablation.txt) — so these modes are asserted at
  the measured envelopes below and REPORTED against the 1e-3 bar in DESIGN.md; fp32 is the parity-grade mode.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_PROB = {"f32": 1e-4, "f16": 1e-2, "bf16": 6e-2}
TOL_MX = 3e-4          # default mode with MX cross-term projections (large forwards): measured 7e-5 worst case at c3; bar 1e-3
TOL_HID = {"f32": 2e-4, "f16": 6e-2, "bf16": 4e-1}
GOLD = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*_b*_s*.npz")))


def sig(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


@pytest.fixture(scope="module")
def engines(weights_for):
    from gliclass.c_amd.engine import Engine
    cache = {}

    def get(cname, dtype):
        if (cname, dtype) not in cache:
            cfg, w = weights_for(cname)
            cache[(cname, dtype)] = Engine(cfg, w, dtype=dtype)
        return cache[(cname, dtype)]
    yield get
    for e in cache.values():
        e.close()


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("case", GOLD)
def test_golden_fixtures(case, dtype, engines, golden_dir):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    cname = str(g["config"])
    if cname == "small" and dtype == "bf16":
        pytest.skip("one 16-bit run of the 141M-parameter case is enough")
    eng = engines(cname, dtype)
    ids, mask = g["ids"].astype(np.int64), g["mask"].astype(np.int64)
    B, S = ids.shape
    eng.keep_hidden(True)
    eng.set_attention_impl(0)
    logits = eng.forward(ids, mask)
    eng.keep_hidden(False)
    assert logits.shape == g["logits"].shape and np.isfinite(logits).all()
    assert eng.last_c == int(g["counts"].max())
    assert np.abs(sig(logits) - g["probs"]).max() <= TOL_PROB[dtype]
    pos, hs = g["sample_pos"], g["hidden_samples"]
    valid = mask[:, pos][:, : hs.shape[2]].astype(bool)
    for which in range(eng.cfg.layers + 1):
        got = eng.hidden(which, B, S)[:, pos, :][..., : hs.shape[-1]][:, : hs.shape[2]]
        assert np.abs(got[valid] - hs[which][valid]).max() <= TOL_HID[dtype], which


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_band_attention_equals_simple_attention(dtype, engines, weights_for):
    """The MFMA Toeplitz-band kernels (2 = one wave per query tile, 3 = workgroup-shared K / V^T ring and p2c image) and the
    straightforward kernel read the same operands, so their layer outputs must agree to accumulation-order noise — including
    S > 512 (clamped buckets), padded lengths that leave waves of the last workgroup without a query tile (Sp = 192, 1152), and
    ragged rows (key tiles skipped past a row's length, padding-only query blocks)."""
    from gliclass.c_amd import synth
    cfg, _ = weights_for("tiny")
    eng = engines("tiny", dtype)
    for (B, S, seed) in ((3, 77, 1), (2, 640, 2), (1, 1100, 3), (4, 150, 4), (2, 1400, 5)):
        ids, mask, _ = synth.make_inputs(cfg, B, S, 2, seed=seed, ragged=True)
        outs = []
        for impl in (1, 2, 3):
            eng.set_attention_impl(impl)
            eng.keep_hidden(True)
            eng.forward(ids, mask)
            outs.append(eng.hidden(1, B, S))
        eng.set_attention_impl(0)
        eng.keep_hidden(False)
        m = mask.astype(bool)
        tol = {"f32": 5e-5, "f16": 2e-2, "bf16": 1.5e-1}[dtype]
        assert np.abs(outs[0][m] - outs[1][m]).max() <= tol, (B, S)
        assert np.abs(outs[0][m] - outs[2][m]).max() <= tol, (B, S)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_live_oracle_sweep(dtype, engines, weights_for):
    """Seeded shapes not in the fixtures: odd S, S not a multiple of 32/64, B not a multiple of anything,
    rows without labels, a fully padded tail row, S=1."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for("mini")
    eng = engines("mini", dtype)
    for (B, S, Cn, lpr, seed) in ((5, 33, 3, [3, 0, 1, 2, 3], 11), (1, 129, 1, None, 12), (7, 64, 2, None, 13), (2, 513, 4, [4, 1], 14)):
        ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=seed, ragged=True, labels_per_row=lpr)
        ref = oracle_c.forward(cfg, w, ids, mask)
        got = eng.forward(ids, mask)
        assert got.shape == ref.shape
        assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB[dtype], (B, S)
    ids = np.array([[cfg.cls_id]], np.int64)
    got = eng.forward(ids, np.ones_like(ids), c_alloc=0)
    assert got.shape == (1, 0)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_last_layer_pruning_is_exact(dtype, engines, weights_for):
    """The default forward computes the last layer only on the rows the head reads; logits must equal the
    unpruned forward (same operands; the selected rows go through the simple attention kernel, so 16-bit
    modes differ by accumulation order only).  Also checks S where the band kernel has saturated tiles."""
    from gliclass.c_amd import synth
    cfg, _ = weights_for("mini")
    eng = engines("mini", dtype)
    for (B, S, lpr, seed) in ((3, 96, [3, 0, 2], 31), (2, 700, None, 32), (1, 1300, None, 33)):
        ids, mask, _ = synth.make_inputs(cfg, B, S, 3, seed=seed, ragged=True, labels_per_row=lpr)
        eng.set_prune_last_layer(True)
        pruned = eng.forward(ids, mask)
        eng.set_prune_last_layer(False)
        full = eng.forward(ids, mask)
        eng.set_prune_last_layer(True)
        tol = 1e-5 if dtype == "f32" else 5e-3
        assert np.abs(sig(pruned) - sig(full)).max() <= tol, (B, S)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_pruned_layer_with_scattered_class_tokens(dtype, engines, weights_for):
    """The pruned last layer computes query tiles only where [CLS] / class tokens sit: the Q third of its QKV GEMM skips 256-row tiles
    without one, and a 4-tile workgroup of the band kernel with exactly ONE selected tile splits that tile's keys over its four waves
    (several selected tiles: one wave per tile as before).  Class tokens are placed by hand: three tiles of the first 128-query group
    (fallback), one tile each in two later groups (cooperative), on rows of different length; pruned == unpruned, and f32 vs the oracle."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for("mini")
    eng = engines("mini", dtype)
    B, S = 3, 640
    ids, mask, _ = synth.make_inputs(cfg, B, S, 0, seed=91, ragged=True, labels_per_row=[0, 0, 0])
    ids[ids == cfg.class_token_index] = 5
    spots = ([1, 40, 70, 200, 300], [2, 150, 500], [300])          # class-token positions per row (row 2: one far tile only besides [CLS])
    for b, ps in enumerate(spots):
        n = int(mask[b].sum())
        for q in ps:
            if q < n - 1:
                ids[b, q] = cfg.class_token_index
    counts = (ids == cfg.class_token_index).sum(1)
    assert counts.min() >= 1
    eng.set_prune_last_layer(True)
    pruned = eng.forward(ids, mask)
    eng.set_prune_last_layer(False)
    full = eng.forward(ids, mask)
    eng.set_prune_last_layer(True)
    C = pruned.shape[1]
    valid = np.arange(C)[None, :] < counts[:, None]
    tol = 1e-5 if dtype == "f32" else 5e-3
    assert np.abs(sig(pruned) - sig(full))[valid].max() <= tol
    if dtype == "f32":
        ref = oracle_c.forward(cfg, w, ids, mask)
        assert np.abs(sig(pruned) - sig(ref))[valid].max() <= TOL_PROB["f32"]


def test_saturated_tiles_match_simple_kernel_long_sequence(engines, weights_for):
    """S = 2048: most key tiles of the band kernel take the constant-delta shortcut (|q-k| beyond the
    bucket clamp); layer output must still match the straightforward kernel."""
    from gliclass.c_amd import synth
    cfg, _ = weights_for("tiny")
    eng = engines("tiny", "f16")
    ids, mask, _ = synth.make_inputs(cfg, 1, 2048, 2, seed=41, ragged=False)
    outs = []
    for impl in (1, 2, 3):
        eng.set_attention_impl(impl)
        eng.keep_hidden(True)
        eng.forward(ids, mask)
        outs.append(eng.hidden(1, 1, 2048))
    eng.set_attention_impl(0)
    eng.keep_hidden(False)
    assert np.abs(outs[0] - outs[1]).max() <= 2e-2
    assert np.abs(outs[0] - outs[2]).max() <= 2e-2


def test_edge_shapes_f32_vs_oracle(engines, weights_for):
    """Edges the reference can produce: MAX_LENGTH-class sequences (S = 4096 > 8 x max_position_embeddings), many
    labels (C = 40), batch sizes that are not multiples of anything, and a row that is padding only."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for("tiny")
    eng = engines("tiny", "f32")
    ids, mask, _ = synth.make_inputs(cfg, 1, 4096, 2, seed=51)
    assert np.abs(sig(eng.forward(ids, mask)) - sig(oracle_c.forward(cfg, w, ids, mask))).max() <= 1e-4
    ids, mask, counts = synth.make_inputs(cfg, 3, 200, 40, seed=52, ragged=True, labels_per_row=[40, 7, 0])
    assert list(counts) == [40, 7, 0]
    got, ref = eng.forward(ids, mask), oracle_c.forward(cfg, w, ids, mask)
    assert got.shape == (3, 40) and np.abs(sig(got) - sig(ref)).max() <= 1e-4
    ids, mask, _ = synth.make_inputs(cfg, 13, 50, 2, seed=53, ragged=True)
    ids[5] = 0
    mask[5] = 0                                              # a batch slot that is padding only
    got, ref = eng.forward(ids, mask), oracle_c.forward(cfg, w, ids, mask)
    keep = np.arange(13) != 5
    assert np.isfinite(got).all() and np.abs(sig(got[keep]) - sig(ref[keep])).max() <= 1e-4
    # f16 band kernel on the same edges: finite, and S = 4096 within the 16-bit envelope
    e16 = engines("tiny", "f16")
    ids, mask, _ = synth.make_inputs(cfg, 1, 4096, 2, seed=51)
    assert np.abs(sig(e16.forward(ids, mask)) - sig(oracle_c.forward(cfg, w, ids, mask))).max() <= TOL_PROB["f16"]


def test_rows_are_independent_and_order_free(engines, weights_for):
    """Size-independent properties at a larger shape: permuting batch rows permutes logits; a row's
    logits do not depend on its batch mates (what makes the batch shard across GPUs)."""
    from gliclass.c_amd import synth
    cfg, _ = weights_for("mini")
    eng = engines("mini", "f16")
    ids, mask, _ = synth.make_inputs(cfg, 16, 512, 4, seed=5, ragged=True)
    base = eng.forward(ids, mask)
    perm = np.random.RandomState(0).permutation(16)
    assert np.array_equal(eng.forward(ids[perm], mask[perm]), base[perm])
    lo = eng.forward(ids[:8], mask[:8])
    hi = eng.forward(ids[8:], mask[8:])
    assert np.array_equal(np.concatenate([lo, hi]), base)
    n = int(mask[3].sum())
    solo = eng.forward(ids[3:4, :n], mask[3:4, :n])          # trimmed to its own length (different Sp)
    assert np.abs(sig(solo) - sig(base[3:4])).max() <= 5e-3


def test_model_h_drop_in_path(weights_for):
    """The reference call sequence (/root/reference/main.c:83-99,141-150 + parallel_processor.c:44,88) through
    include/model.h: initialize_ort_api -> env -> session -> prepare_input_tensors -> run_inference ->
    OrtApi introspection -> ReleaseValue, checked against the oracle."""
    import oracle_c
    from gliclass.c_amd import _lib, synth
    cfg, w = weights_for("tiny")
    m = _lib.model()
    os.environ["GLICLASS_DTYPE"] = "f32"
    m.initialize_ort_api()
    env = m.initialize_ort_environment()
    sess = m.create_ort_session(env, b"synthetic:tiny:42", 8)
    assert sess and m.glc_session_num_devices(sess) == 1
    ids, mask, _ = synth.make_inputs(cfg, 3, 50, 3, seed=21, ragged=True, labels_per_row=[3, 1, 2])
    i32, m32 = ids.astype(np.int32), mask.astype(np.int32)
    rows_i = (C.POINTER(C.c_int) * 3)(*[i32[b].ctypes.data_as(C.POINTER(C.c_int)) for b in range(3)])
    rows_m = (C.POINTER(C.c_int) * 3)(*[m32[b].ctypes.data_as(C.POINTER(C.c_int)) for b in range(3)])
    tok = _lib.TokenizedInputs(rows_i, rows_i, rows_m, 3, 50)
    a, b = C.POINTER(_lib.OrtValue)(), C.POINTER(_lib.OrtValue)()
    assert m.prepare_input_tensors(C.byref(tok), C.byref(a), C.byref(b)) == 0
    out = m.run_inference(sess, a, b)
    assert out and out.contents.type == 1 and list(out.contents.dims[:2]) == [3, 3]
    got = np.ctypeslib.as_array(C.cast(out.contents.data, C.POINTER(C.c_float)), shape=(3, 3)).copy()
    ref = oracle_c.forward(cfg, w, ids, mask)
    assert np.abs(sig(got) - sig(ref)).max() <= 1e-4
    # batch sharding extension: 3 batches through parallel_inference give the same tensors
    ins_a = (C.POINTER(_lib.OrtValue) * 3)(a, a, a)
    ins_b = (C.POINTER(_lib.OrtValue) * 3)(b, b, b)
    outs = (C.POINTER(_lib.OrtValue) * 3)()
    m.parallel_inference(sess, ins_a, ins_b, 3, outs)
    for o in outs:
        assert o and np.array_equal(np.ctypeslib.as_array(C.cast(o.contents.data, C.POINTER(C.c_float)), shape=(3, 3)), got)
    del os.environ["GLICLASS_DTYPE"]


def test_pure_c_driver_end_to_end(tmp_path, weights_for):
    """examples/run_pretokenized.c (plain C, the reference's main.c sequence): 11 ragged rows -> 2 batches of <= 8
    (BATCH_SIZE chunking + short last chunk), printed scores must equal sigmoid(oracle logits)."""
    import re
    import subprocess
    import oracle_c
    from gliclass.c_amd import synth
    from test_host import _build_example
    cfg, w = weights_for("tiny")
    exe = _build_example(tmp_path)
    ids, mask, _ = synth.make_inputs(cfg, 11, 70, 2, seed=17, ragged=True)
    tok = tmp_path / "tok.txt"
    tok.write_text("\n".join(" ".join(str(int(t)) for t in ids[b, : int(mask[b].sum())]) for b in range(11)) + "\n")
    env = dict(os.environ, GLICLASS_DTYPE="f32", GLICLASS_THRESHOLD="0.0")
    r = subprocess.run([exe, "synthetic:tiny:42", str(tok), "multi-label", "alpha", "beta"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    got = {}
    batch_of_line = 0
    for line in r.stdout.splitlines():
        m = re.match(r"  Text_(\d+) Label: (\w+), Score: ([0-9.]+)", line)
        if m:
            got.setdefault((int(m.group(1)), m.group(2)), []).append(float(m.group(3)))
    # every (local text index, label) appears once per batch that has that index: indices 0-2 twice (batches of 8 and 3)
    assert len(got[(0, "alpha")]) == 2 and len(got[(7, "alpha")]) == 1
    scores = sorted(v for vs in got.values() for v in vs)
    ref = []
    for lo, hi in ((0, 8), (8, 11)):
        S = int(mask[lo:hi].sum(1).max())                     # pad-to-longest inside each batch
        lg = oracle_c.forward(cfg, w, ids[lo:hi, :S], mask[lo:hi, :S])
        ref += list(sig(lg).ravel())
    assert len(scores) == len(ref) == 22
    assert np.abs(np.array(scores) - np.array(sorted(ref))).max() <= 2e-5      # %.6f printing + fp32 noise


@pytest.mark.parametrize("cname,prefix", [("mini", "encoder_model.model."), ("dec-mini", "decoder_model.model.")])
def test_checkpoint_import_path_vs_live_hf(cname, prefix, tmp_path):
    """an HF DebertaV2Model / Qwen2Model state_dict (+ head tensors) -> weights.from_state_dict -> .glcw blob ->
    create_ort_session(path) -> run_inference, compared with the live HF forward of the SAME module."""
    transformers = pytest.importorskip("transformers")
    import torch
    import hf_ref
    from gliclass.c_amd import _lib, synth, weights
    from gliclass.c_amd.config import CONFIGS
    cfg = CONFIGS[cname]
    base = weights.make_weights(cfg, 5)
    torch.manual_seed(0)
    model = hf_ref.build_hf_model(cfg, base)
    with torch.no_grad():                                  # perturb so the blob really comes from the module
        for p in model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    sd = {prefix + k: v for k, v in model.state_dict().items()}   # gliclass-style prefix
    sd.update({k: torch.from_numpy(v) for k, v in base.items() if "projector" in k})
    tensors = weights.from_state_dict(sd, cfg)
    path = str(tmp_path / (cname + ".glcw"))
    weights.write_blob(path, cfg, tensors)
    ids, mask, _ = synth.make_inputs(cfg, 3, 90, 3, seed=9, ragged=True)
    ref = hf_ref.forward(cfg, tensors, ids, mask, model=model)
    m = _lib.model()
    os.environ["GLICLASS_DTYPE"] = "f32"
    m.initialize_ort_api()
    sess = m.create_ort_session(m.initialize_ort_environment(), path.encode(), 8)
    del os.environ["GLICLASS_DTYPE"]
    assert sess
    i64, m64 = np.ascontiguousarray(ids, np.int64), np.ascontiguousarray(mask, np.int64)
    a = m.create_tensor(i64.ctypes.data, 3, 90)
    b = m.create_tensor(m64.ctypes.data, 3, 90)
    out = m.run_inference(sess, a, b)
    assert out and list(out.contents.dims[:2]) == [3, 3]
    got = np.ctypeslib.as_array(C.cast(out.contents.data, C.POINTER(C.c_float)), shape=(3, 3)).copy()
    assert np.abs(sig(got) - sig(ref)).max() <= 1e-4


def test_full_size_base_row_vs_oracle(weights_for):
    """BASELINE config c3's model (gliclass-base shape) at S=1024: one row against the fp32 CPU oracle,
    plus batch-position invariance at B=8."""
    import oracle_c
    from gliclass.c_amd import synth
    from gliclass.c_amd.engine import Engine
    cfg, w = weights_for("base")
    ids, mask, _ = synth.make_inputs(cfg, 8, 1024, 8, seed=1234)
    ref = oracle_c.forward(cfg, w, ids[:1], mask[:1])
    res = {}
    for dtype in ("f32", "f16"):
        eng = Engine(cfg, w, dtype=dtype)
        got = eng.forward(ids, mask) if dtype == "f16" else eng.forward(ids[:2], mask[:2])
        eng.close()
        assert np.isfinite(got).all()
        res[dtype] = float(np.abs(sig(got[:1]) - sig(ref)).max())
    print("base S=1024 max prob err vs oracle:", res)
    assert res["f32"] <= 1e-4
    assert res["f16"] <= 1e-2          # measured 0.6e-3 ... 2.7e-3 (12 layers)


def test_large_config_c4_shape(c_generated_weights):
    """BASELINE.json configs[3] (gliclass-large shape: 24 layers, H 1024, 16 heads, I 4096) — the per-GPU shard of the 8-GPU run is
    the same engine at B = 32; here a short ragged batch against the oracle in the parity-grade mode, plus f16 inside its envelope."""
    import oracle_c
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.engine import Engine
    cfg = CONFIGS["large"]
    w = c_generated_weights("synthetic:large:42", cfg)
    ids, mask, _ = synth.make_inputs(cfg, 2, 200, 4, seed=41, ragged=True, labels_per_row=[4, 2])
    ref = oracle_c.forward(cfg, w, ids, mask)
    for dtype in ("f32", "f16"):
        eng = Engine.from_spec(cfg, "synthetic:large:42", dtype=dtype)
        try:
            got = eng.forward(ids, mask)
        finally:
            eng.close()
        assert got.shape == ref.shape
        assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB[dtype], dtype


@pytest.mark.parametrize("pooling", ["avg", "last"])
def test_pooling_switches(pooling, weights_for):
    """pooling = 'avg' (mean over the attended positions) and 'last' are config switches of the head (SURVEY.md §8a row a12:
    "must be config-switchable"; upstream semantics unpinned); checked against the oracle in fp32 and f16.  These switches
    disable the last-layer pruning (it assumes the pooled row is position 0)."""
    import dataclasses
    import oracle_c
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.config import POOL_AVG, POOL_LAST
    from gliclass.c_amd.engine import Engine
    base, _ = weights_for("tiny")
    cfg = dataclasses.replace(base, pooling={"avg": POOL_AVG, "last": POOL_LAST}[pooling])
    w = weights.make_weights(cfg, 42)
    ids, mask, _ = synth.make_inputs(cfg, 4, 150, 3, seed=17, ragged=True, labels_per_row=[3, 1, 0, 2])
    ref = oracle_c.forward(cfg, w, ids, mask)
    for dtype in ("f32", "f16"):
        eng = Engine(cfg, w, dtype=dtype)
        try:
            got = eng.forward(ids, mask)
        finally:
            eng.close()
        assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB[dtype], dtype


@pytest.mark.parametrize("scorer", ["weighted-dot", "mlp"])
def test_scorer_switches(scorer, weights_for):
    """scorer_type = 'weighted-dot' / 'mlp' ("must be config-switchable"; module structure restated from
    the upstream package, include/gliclass_hip.h — parity with upstream unpinned; the oracle's versions are pinned to torch.nn modules in
    tests/test_oracle.py).  Engine vs oracle in fp32 and f16, ragged rows with different label counts (padded class slots score the
    projected zero row, as in the oracle); the encoder through Python-side tensors, the decoder backbone and the pruned path through
    the C weight source's "synthetic:<config>:<seed>:<scorer>" spec (same PRNG, so the same tensors)."""
    import dataclasses
    import oracle_c
    from gliclass.c_amd import synth, weights
    from gliclass.c_amd.config import CONFIGS, SCORER_NAMES
    from gliclass.c_amd.engine import Engine
    for cname, shape in (("tiny", (4, 150, 3, [3, 1, 0, 2])), ("dec-tiny", (3, 96, 2, [2, 1, 2])), ("mini", (5, 200, 4, [4, 4, 2, 0, 3]))):
        cfg = dataclasses.replace(CONFIGS[cname], scorer=SCORER_NAMES[scorer])
        w = weights.make_weights(cfg, 42)
        B, S, Cn, per_row = shape
        ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=17, ragged=True, labels_per_row=per_row)
        ref = oracle_c.forward(cfg, w, ids, mask)
        assert np.abs(ref).max() > 0.05
        for dtype in ("f32", "f16"):
            eng = Engine(cfg, w, dtype=dtype) if cname == "tiny" else Engine.from_spec(cfg, f"synthetic:{cname}:42:{scorer}", dtype=dtype)
            try:
                got = eng.forward(ids, mask)
            finally:
                eng.close()
            assert got.shape == ref.shape
            assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB[dtype], (cname, dtype)
    # normalize_features with these scorers: features L2-normalised before the scorer GEMMs, logits times logit_scale
    cfg = dataclasses.replace(CONFIGS["tiny"], scorer=SCORER_NAMES[scorer], normalize_features=True, logit_scale=3.5)
    w = weights.make_weights(cfg, 42)
    ids, mask, _ = synth.make_inputs(cfg, 4, 150, 3, seed=17, ragged=True, labels_per_row=[3, 1, 0, 2])
    ref = oracle_c.forward(cfg, w, ids, mask)
    plain = oracle_c.forward(dataclasses.replace(cfg, normalize_features=False, logit_scale=1.0), w, ids, mask)
    assert np.abs(ref - plain).max() > 1e-3, "the switch must change the logits"
    eng = Engine(cfg, w, dtype="f32")
    try:
        got = eng.forward(ids, mask)
    finally:
        eng.close()
    assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB["f32"]


def test_length_bucketing_decoder_backbone(engines, weights_for):
    """The same bucketing on the decoder backbone (last-token pooling reads each group's own lengths)."""
    from gliclass.c_amd import synth
    cfg, _ = weights_for("dec-mini")
    eng = engines("dec-mini", "f32")
    B, S = 400, 640                                    # 8 waves of 256-row tiles as one padded batch (hidden 512), ~6 when grouped
    ids, mask, _ = synth.make_inputs(cfg, B, S, 2, seed=78, ragged=True)
    eng.set_length_buckets(1)
    one = eng.forward(ids, mask)
    eng.set_length_buckets(4)
    got = eng.forward(ids, mask)
    assert eng.L.glc_debug_last_forward_groups(eng.h) >= 2
    assert np.abs(sig(got) - sig(one)).max() <= 1e-5


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_length_bucketing_preserves_results(dtype, engines, weights_for):
    """glc_engine_forward splits a ragged batch into length groups (default 4); every row's logits must equal the single padded
    batch (rows are independent; columns past the last attended token contribute nothing).  Includes a row without labels and
    a full-length row."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for("small")                      # hidden 768: a wave of 256-row tiles is 21 845 rows, so this batch is 6 waves
    eng = engines("small", dtype)
    B, S = 256, 512
    ids, mask, _ = synth.make_inputs(cfg, B, S, 3, seed=77, ragged=True, labels_per_row=[3, 0, 1, 2] * 64)
    mask[0, :] = 1                                     # one full-length row
    lens = mask.sum(1)
    assert lens.min() < 0.6 * S                        # really ragged
    eng.set_length_buckets(1)
    one = eng.forward(ids, mask)
    assert eng.L.glc_debug_last_forward_groups(eng.h) == 1
    eng.set_length_buckets(4)
    got = eng.forward(ids, mask)
    assert eng.L.glc_debug_last_forward_groups(eng.h) >= 2          # the plan really split this batch
    assert got.shape == one.shape and eng.last_c == 3
    # f32: the groups are small enough for the plain-fp32 128-tile pipeline while the single batch runs the group-split one with
    # LayerNorm folded into its GEMMs — same function, different rounding points, each ~1e-5 from the oracle (checked below);
    # 16-bit: a different padded length moves tile boundaries (rounding only)
    # (round 3: forwards large enough for the 256-tile pipeline run the MX cross-term arithmetic, ~1e-4 from the oracle — TOL_MX)
    tol = TOL_MX if dtype == "f32" else 5e-3
    assert np.abs(sig(got) - sig(one)).max() <= tol
    if dtype == "f32":                                 # and against the oracle on a few rows (trimmed: rows are independent)
        for b in (0, 2, 19):                          # rows with 3, 1 and 2 labels
            n = int(mask[b].sum())
            ref = oracle_c.forward(cfg, w, ids[b:b + 1, :n], mask[b:b + 1, :n])
            k = ref.shape[1]
            assert np.abs(sig(got[b, :k]) - sig(ref[0])).max() <= TOL_MX


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cname", ["tiny", "dec-tiny"])
def test_randomised_shape_sweep(cname, dtype, engines, weights_for):
    """Seeded random shapes for both backbones, in the parity-grade mode and in f16 (the MFMA attention kernels): B in 1..9, S in 1..700 (any residue mod 32/64), 0..6
    labels per row, ragged lengths incl. rows that are almost empty, and a mask hole inside a row (attended tokens after a
    masked one).  Each case is checked against the oracle; a failure prints the case so it can be replayed."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for(cname)
    eng = engines(cname, dtype)
    rng = np.random.default_rng(20261003)
    for case in range(24):
        B = int(rng.integers(1, 10))
        S = int(rng.integers(1, 701)) if case % 6 else int(rng.integers(1, 40))
        Cmax = int(rng.integers(0, 7))
        lpr = [int(x) for x in rng.integers(0, Cmax + 1, size=B)]
        need = 2 + 3 * max(lpr + [0]) + 2
        if S < need:
            S = need
        ids, mask, counts = synth.make_inputs(cfg, B, S, max(Cmax, 1), seed=1000 + case, ragged=bool(case % 2), labels_per_row=lpr)
        if case % 5 == 0 and S > 20:                       # a hole: masked tokens in the middle of row 0
            n0 = int(mask[0].sum())
            if n0 > 16:
                mask[0, n0 // 2: n0 // 2 + 3] = 0
        ref = oracle_c.forward(cfg, w, ids, mask)
        got = eng.forward(ids, mask)
        assert got.shape == ref.shape, (case, B, S, lpr)
        if ref.size:
            assert np.abs(sig(got) - sig(ref)).max() <= TOL_PROB[dtype], (case, B, S, lpr)


@pytest.mark.parametrize("gemm,attn", [("native", "native"), ("split", "native"), ("native", "split")])
def test_fp32_mode_kernel_choices_agree(gemm, attn, tmp_path):
    """The fp32 mode's split-f16 products (default) and the plain fp32-MFMA kernels (GLICLASS_F32_GEMM / GLICLASS_F32_ATTN = native)
    are interchangeable: every combination stays inside the f32 bound against the oracle.  (Own process: the switches are read once.)"""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'oracle')!r}]\n"
        "import oracle_c\n"
        "from gliclass.c_amd import synth, weights\n"
        "from gliclass.c_amd.config import CONFIGS\n"
        "from gliclass.c_amd.engine import Engine\n"
        "cfg = CONFIGS['mini']; w = weights.make_weights(cfg, 42)\n"
        "ids, mask, _ = synth.make_inputs(cfg, 3, 700, 4, seed=5, ragged=True)\n"
        "ref = oracle_c.forward(cfg, w, ids, mask)\n"
        "got = Engine(cfg, w, dtype='f32').forward(ids, mask)\n"
        "sig = lambda x: 1 / (1 + np.exp(-x.astype(np.float64)))\n"
        "print('ERR', float(np.abs(sig(got) - sig(ref)).max()))\n")
    env = dict(os.environ, GLICLASS_F32_GEMM=gemm, GLICLASS_F32_ATTN=attn)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    err = float(r.stdout.strip().split("ERR")[-1])
    assert err <= 1e-4, (gemm, attn, err)


def test_masked_class_token_is_refused(engines, weights_for):
    """A <<LABEL>> token under attention_mask 0 would need a padding-QUERY row (HF: uniform attention over every position); the engine
    does not compute those, so the host-buffer forward refuses the input instead of returning a different number."""
    from gliclass.c_amd import synth
    cfg, _ = weights_for("tiny")
    eng = engines("tiny", "f32")
    ids, mask, _ = synth.make_inputs(cfg, 2, 64, 3, seed=1)
    pos = int(np.where(ids[0] == cfg.class_token_index)[0][1])
    mask[0, pos] = 0
    with pytest.raises(RuntimeError, match="class token"):
        eng.forward(ids, mask)
    mask[0, pos] = 1
    assert np.isfinite(eng.forward(ids, mask)).all()


def test_out_of_range_activation_fails_loudly(weights_for, tmp_path):
    """Every matrix product runs on f16 MFMA operands (the default fp32 mode as split-f16 pairs), so an activation beyond 65504
    becomes inf / NaN inside the kernels.  The host-buffer forward must refuse to return such logits ("a per-forward
    overflow flag ... fail or fall back"); the fp32-MFMA kernels (GLICLASS_F32_GEMM/ATTN=native) handle the same model."""
    import subprocess
    import sys
    from gliclass.c_amd import synth
    from gliclass.c_amd.engine import Engine
    cfg, w0 = weights_for("tiny")
    w = dict(w0)
    w["embeddings.LayerNorm.weight"] = w0["embeddings.LayerNorm.weight"] * 3.0e5          # hidden states of the first layer ~ 3e5
    ids, mask, _ = synth.make_inputs(cfg, 2, 80, 3, seed=3, ragged=True)
    for dtype in ("f32", "f16"):
        eng = Engine(cfg, w, dtype=dtype)
        try:
            with pytest.raises(RuntimeError, match="non-finite logit"):
                eng.forward(ids, mask)
        finally:
            eng.close()
    code = (
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{ROOT!r}]\n"
        "from gliclass.c_amd import synth, weights\n"
        "from gliclass.c_amd.config import CONFIGS\n"
        "from gliclass.c_amd.engine import Engine\n"
        "cfg = CONFIGS['tiny']; w = weights.make_weights(cfg, 42)\n"
        "w['embeddings.LayerNorm.weight'] = w['embeddings.LayerNorm.weight'] * 3.0e5\n"
        "ids, mask, _ = synth.make_inputs(cfg, 2, 80, 3, seed=3, ragged=True)\n"
        "got = Engine(cfg, w, dtype='f32').forward(ids, mask)\n"
        "print('FINITE', bool(np.isfinite(got).all()))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GLICLASS_F32_GEMM="native", GLICLASS_F32_ATTN="native"))
    assert r.returncode == 0 and "FINITE True" in r.stdout, r.stderr[-2000:]


def test_second_device_gives_identical_results(weights_for):
    """An engine on GPU 1 (when the box has one) runs the same kernels as an engine on GPU 0: bit-identical logits — the
    property the batch shard over GPUs rests on.  Also covers the per-device CU-count cache and LDS-limit mask."""
    from gliclass.c_amd import _lib, synth
    from gliclass.c_amd.engine import Engine
    if _lib.hip().glc_device_count() < 2:
        pytest.skip("one GPU visible")
    cfg, w = weights_for("mini")
    ids, mask, _ = synth.make_inputs(cfg, 9, 300, 4, seed=8, ragged=True)
    for dtype in ("f32", "f16"):
        e0, e1 = Engine(cfg, w, dtype=dtype, device=0), Engine(cfg, w, dtype=dtype, device=1)
        try:
            a, b = e0.forward(ids, mask), e1.forward(ids, mask)
            assert np.array_equal(a, b), dtype
        finally:
            e0.close(); e1.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_layernorm_fold_16bit_modes(dtype, engines, weights_for):
    """16-bit modes, forwards large enough for the staggered 256-tile GEMM: LayerNorm folded into the GEMMs around it (raw rows of T +
    row statistics out of the residual GEMMs, gamma-folded weights + epilogue correction in QKV / FFN1) against LayerNorm as kernels of
    its own — same function, other rounding points — and a few rows against the oracle."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for("small")
    eng = engines("small", dtype)
    B, S = 40, 512
    ids, mask, _ = synth.make_inputs(cfg, B, S, 4, seed=123, ragged=True)
    eng.set_length_buckets(1)
    try:
        fused = eng.forward(ids, mask)
        assert eng.last_ln_folded(), "this shape should run the folded pipeline"
        eng.set_ln_fused(False)
        unf = eng.forward(ids, mask)
        assert not eng.last_ln_folded()
    finally:
        eng.set_ln_fused(True)
        eng.set_length_buckets(4)
    assert np.isfinite(fused).all() and not np.array_equal(fused, unf)
    tol = TOL_MX if dtype == "f32" else TOL_PROB[dtype]      # f32: the folded forward of this shape runs the MX arithmetic, the unfused one split-f16
    assert np.abs(sig(fused) - sig(unf)).max() <= tol
    for b in (0, 7, 33):
        n = int(mask[b].sum())
        ref = oracle_c.forward(cfg, w, ids[b:b + 1, :n], mask[b:b + 1, :n])
        assert np.abs(sig(fused[b:b + 1, :ref.shape[1]]) - sig(ref)).max() <= tol, b


@pytest.mark.parametrize("cname", ["mini", "small"])
def test_group_split_pipeline_vs_oracle_and_plain_fp32(cname, engines, weights_for):
    """fp32 mode, the two activation formats: plain fp32 rows + 128-tile split-f16 GEMMs (small forwards) and group-split rows
    ([32 hi | 32 lo] f16 groups) + the 256-tile LDS-DMA GEMM as a K' = 3K loop (large forwards; forced here).  Same arithmetic
    (three f16 MFMAs per product, fp32 accumulation), different tiling and summation order: both against the oracle, and against
    each other far inside the f32 bound.  Ragged rows, S that is no multiple of 64, rows without labels."""
    import oracle_c
    from gliclass.c_amd import synth
    cfg, w = weights_for(cname)
    eng = engines(cname, "f32")
    shapes = ((3, 200, 3, [3, 0, 2], 61), (2, 700, 4, None, 62), (9, 64, 2, None, 63), (1, 1100, 5, None, 64)) if cname == "mini" else ((4, 300, 4, [4, 1, 0, 3], 65),)
    try:
        for (B, S, Cn, lpr, seed) in shapes:
            ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=seed, ragged=True, labels_per_row=lpr)
            ref = oracle_c.forward(cfg, w, ids, mask)
            eng.set_group_split(0)
            plain = eng.forward(ids, mask)
            assert not eng.last_group_split()
            eng.set_group_split(2)
            eng.set_mx(False)                 # the split-f16 projections (three f16 MFMAs per product); the MX pipeline has its own test below
            gs = eng.forward(ids, mask)
            assert eng.last_group_split() and not eng.last_mx()
            assert np.isfinite(gs).all()
            assert np.abs(sig(gs) - sig(ref)).max() <= TOL_PROB["f32"], (B, S)
            # ... and with LayerNorm as kernels of its own instead of folded into the GEMMs around it (raw rows + row statistics
            # out of the producer, gamma-folded weights + epilogue correction in the consumer): same function, other rounding points
            eng.set_ln_fused(False)
            unf = eng.forward(ids, mask)
            eng.set_ln_fused(True)
            assert eng.last_group_split()
            assert np.abs(sig(unf) - sig(ref)).max() <= TOL_PROB["f32"], (B, S)
            assert np.abs(sig(gs) - sig(unf)).max() <= 1e-4, (B, S)
            assert not np.array_equal(gs, unf), "the LayerNorm switch changed nothing: is the folded path running?"
            assert np.abs(sig(gs) - sig(plain)).max() <= 1e-4, (B, S)     # each sits ~1e-5 from the oracle; GS rows also carry 22 instead of 24 bits
            # MX cross-term pipeline (the default of large forwards: a_hi*w_hi in f16 MFMAs, both cross terms in one block-scaled fp8
            # MFMA, GX rows): against the oracle inside a third of the 1e-3 bar, and against the split pipeline
            eng.set_mx(True)
            mx = eng.forward(ids, mask)
            assert eng.last_group_split() and eng.last_mx(), "the MX pipeline did not run"
            assert np.isfinite(mx).all()
            assert np.abs(sig(mx) - sig(ref)).max() <= TOL_MX, (B, S)
            assert np.abs(sig(mx) - sig(gs)).max() <= TOL_MX, (B, S)
            assert not np.array_equal(mx, gs)
            # ... and its attention on the bucket-space kernel (round 4, attention_mx2.hip; opt-in): short rows, rows longer than the table's
            # linear range (S = 700, 1100: log buckets and saturation), a single query tile (S = 64), ragged lengths, rows without labels
            eng.set_mx2(True)
            mx2 = eng.forward(ids, mask)
            eng.set_mx2(False)
            assert eng.last_mx_attention() and np.isfinite(mx2).all()
            assert np.abs(sig(mx2) - sig(ref)).max() <= TOL_MX, (B, S)
            assert np.abs(sig(mx2) - sig(mx)).max() <= 1e-4, (B, S)
    finally:
        eng.set_group_split(1)
        eng.set_ln_fused(True)
        eng.set_mx(True)
        eng.set_mx2(False)
