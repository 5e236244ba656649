"""CPU (no GPU): the C-ABI libraries load and export every declared symbol; host-side logic of the
drop-in layer (tensor helpers, OrtApi shim, post-processor, weight source, PRNG, delta table)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def libs():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gliclass", "c_amd"), "-j4", "all"], stdout=subprocess.DEVNULL)
    from gliclass.c_amd import _lib
    return _lib, _lib.hip(), _lib.model()


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct OrtApi \{.*?\} OrtApi;", "", src, flags=re.S)
    src = re.sub(r"typedef struct OrtApiBase \{.*?\} OrtApiBase;", "", src, flags=re.S)
    src = re.sub(r"static inline[^{]*\{[^}]*\}", "", src)
    return set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src))


def test_every_declared_symbol_is_exported(libs):
    _lib, hip, model = libs
    for name in _declared_functions("gliclass_hip.h"):
        assert hasattr(hip, name), name
    for hdr in ("model.h", "postprocessor.h", "parallel_processor.h", "glc_weights.h"):
        for name in _declared_functions(hdr):
            assert hasattr(model, name), (hdr, name)
    assert hasattr(model, "OrtGetApiBase") and hasattr(model, "g_ort")
    # externally supplied pieces (reference's src/preprocessor.c / src/tokenizer.c) are weak, not exported
    for name in _declared_functions("preprocessor.h"):
        assert hasattr(model, name), name
    # the native tokenizer (tokenizers-cpp's C API + the reference's src/tokenizer.c surface) and the JSON front-end
    for hdr in ("tokenizer.h", "tokenizers_c.h", "read_data.h"):
        for name in _declared_functions(hdr):
            assert hasattr(model, name), (hdr, name)
    assert _declared_functions("tokenizer.h") == {"tokenize_inputs", "print_tokenized_inputs", "free_tokenized_inputs", "create_tokenizer"}


def test_no_gpu_means_loud_failure(libs):
    _lib, hip, model = libs
    if hip.glc_device_count() > 0:
        pytest.skip("GPU present")
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd import weights
    from gliclass.c_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no HIP device"):
        Engine(CONFIGS["tiny"], weights.make_weights(CONFIGS["tiny"], 1), dtype="f32")
    model.initialize_ort_api()
    env = model.initialize_ort_environment()
    assert env
    assert not model.create_ort_session(env, b"synthetic:tiny", 8)     # NULL + message on stderr


def test_delta_table_product_impl_bit_exact(libs, golden_dir):
    from gliclass.c_amd.engine import delta_table
    tabs = np.load(os.path.join(golden_dir, "delta_tables.npz"))
    for key in tabs.files:
        assert np.array_equal(delta_table(int(key[1:])), tabs[key].astype(np.int32)), key


def test_prng_and_tensor_specs_match_python(libs):
    _lib, hip, model = libs
    from gliclass.c_amd import prng, weights
    from gliclass.c_amd.config import CONFIGS
    for name, n, amp, mean in (("embeddings.word_embeddings.weight", 5000, 1.0, 0.0), ("x.bias", 17, 0.1, 1.0)):
        out = np.zeros(n, np.float32)
        model.glc_prng_fill(42, name.encode(), n, amp, mean, out.ctypes.data)
        assert np.array_equal(out, prng.uniform_f32(42, name, n, amp, mean))
        assert model.glc_fnv1a64(name.encode()) == prng.fnv1a64(name)
    for cname in ("tiny", "mini", "small", "base", "large", "dec-tiny", "dec-mini", "qwen-1.5b"):
        cfg = CONFIGS[cname]
        cc = _lib.ModelConfig()
        assert model.glc_named_config(cname.encode(), C.byref(cc)) == 0
        assert abs(cc.ln_eps - cfg.ln_eps) < 1e-12 and abs(cc.rope_theta - cfg.rope_theta) < 1.0
        for f in ("vocab", "hidden", "layers", "heads", "head_dim", "inter", "class_token_index", "text_token_index", "pos_buckets",
                  "backbone", "kv_heads", "causal", "pooling"):
            assert getattr(cc, f) == getattr(cfg, f), (cname, f)
        specs = weights.tensor_specs(cfg)
        buf = C.create_string_buffer(96)
        shp = (C.c_uint64 * 4)()
        amp, mean = C.c_double(), C.c_double()
        for i, (n, shape, a, m) in enumerate(specs):
            nd = model.glc_tensor_spec(C.byref(cc), i, buf, shp, C.byref(amp), C.byref(mean))
            assert nd == len(shape) and buf.value.decode() == n and tuple(shp[:nd]) == tuple(shape)
            assert abs(amp.value - a) < 1e-15 and mean.value == m
        assert model.glc_tensor_spec(C.byref(cc), len(specs), buf, shp, C.byref(amp), C.byref(mean)) == -1


def test_weight_sources_synthetic_and_blob(libs, tmp_path):
    _lib, hip, model = libs
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    import dataclasses
    from gliclass.c_amd.config import SCORER_NAMES
    for cname, scorer in (("tiny", None), ("dec-tiny", None), ("tiny", "weighted-dot"), ("dec-tiny", "mlp")):
        # both backbone families (the blob header, v2, carries the family) and the scorers that bring tensors of their own
        cfg = CONFIGS[cname] if scorer is None else dataclasses.replace(CONFIGS[cname], scorer=SCORER_NAMES[scorer])
        ref = weights.make_weights(cfg, 7)
        names = [s[0] for s in weights.tensor_specs(cfg)]
        path = str(tmp_path / (cname + ".glcw"))
        weights.write_blob(path, cfg, ref)
        cfg2, back = weights.read_blob(path)
        assert cfg2.hidden == cfg.hidden and cfg2.backbone == cfg.backbone and cfg2.kv_heads == cfg.kv_heads
        assert cfg2.causal == cfg.causal and abs(cfg2.rope_theta - cfg.rope_theta) < 1.0 and cfg2.pooling == cfg.pooling
        assert all(np.array_equal(back[n], ref[n]) for n in names) and cfg2.scorer == cfg.scorer
        for src in ((f"synthetic:{cname}:7" + (":" + scorer if scorer else "")).encode(), path.encode()):
            W = _lib.Weights()
            assert model.glc_weights_load(src, C.byref(W)) == 0
            assert W.n_tensors == len(names) and W.cfg.hidden == cfg.hidden and W.cfg.class_token_index == cfg.class_token_index
            assert W.cfg.backbone == cfg.backbone and W.cfg.kv_heads == cfg.kv_heads and W.cfg.head_dim == cfg.head_dim
            assert W.cfg.scorer == cfg.scorer
            for i, n in enumerate(names):
                got = np.ctypeslib.as_array(W.tensors[i], shape=(ref[n].size,))
                assert np.array_equal(got, ref[n].ravel()), n
            model.glc_weights_free(C.byref(W))
    W = _lib.Weights()
    assert model.glc_weights_load(b"/nonexistent.glcw", C.byref(W)) != 0
    assert model.glc_weights_load(b"synthetic:nope", C.byref(W)) != 0
    assert model.glc_weights_load(b"synthetic:tiny:7:hopfield", C.byref(W)) != 0       # a scorer that is not implemented is refused by name
    bad = tmp_path / "bad.glcw"
    bad.write_bytes(b"\0" * 512)
    assert model.glc_weights_load(str(bad).encode(), C.byref(W)) != 0


def test_prompt_builder_matches_reference_strings(libs):
    """/root/reference/src/preprocessor.c:67-111: "<<LABEL>>" + lower(label) ..., "<<SEP>>", prompt_first order."""
    _lib, hip, model = libs

    def one(text, labels, prompt_first):
        arr = (C.c_char_p * max(len(labels), 1))(*[l.encode() for l in labels])
        p = model.prepare_input(text.encode(), arr, len(labels), prompt_first)
        out = C.string_at(p).decode()
        C.CDLL(None).free(C.c_void_p(p))
        return out
    assert one("Some Text.", ["Format", "MODEL x"], True) == "<<LABEL>>format<<LABEL>>model x<<SEP>>Some Text."
    assert one("Some Text.", ["Format", "MODEL x"], False) == "Some Text.<<LABEL>>format<<LABEL>>model x<<SEP>>"
    assert one("t", [], True) == "<<SEP>>t" and one("", ["A"], False) == "<<LABEL>>a<<SEP>>"
    texts = (C.c_char_p * 2)(b"one", b"two")
    l0 = (C.c_char_p * 2)(b"X", b"y")
    l1 = (C.c_char_p * 1)(b"Zed")
    labs = (C.POINTER(C.c_char_p) * 2)(C.cast(l0, C.POINTER(C.c_char_p)), C.cast(l1, C.POINTER(C.c_char_p)))
    nl = (C.c_size_t * 2)(2, 1)
    out = model.prepare_inputs(texts, labs, 2, nl, False, True)
    assert [C.string_at(out[i]).decode() for i in range(2)] == ["<<LABEL>>x<<LABEL>>y<<SEP>>one", "<<LABEL>>zed<<SEP>>two"]
    model.free_prepared_inputs(out, 2)
    out = model.prepare_inputs(texts, labs, 2, nl, True, False)        # same_labels: slot 0 for every text
    assert [C.string_at(out[i]).decode() for i in range(2)] == ["one<<LABEL>>x<<LABEL>>y<<SEP>>", "two<<LABEL>>x<<LABEL>>y<<SEP>>"]
    model.free_prepared_inputs(out, 2)


def _ragged_rows(rows):
    arrs = [np.asarray(r, np.int32) for r in rows]
    ptrs = (C.POINTER(C.c_int) * len(arrs))(*[a.ctypes.data_as(C.POINTER(C.c_int)) for a in arrs])
    return arrs, ptrs


def test_flatten_create_prepare_tensors(libs):
    """/root/reference/src/model.c:17-108 semantics, incl. ownership and shapes."""
    _lib, hip, model = libs
    model.initialize_ort_api()
    rows = [[1, 5, 7, 0], [1, 9, 0, 0], [-3, 2**31 - 1, 4, 4]]
    arrs, ptrs = _ragged_rows(rows)
    flat = model.flatten_int_array(ptrs, 3, 4)
    got = np.ctypeslib.as_array(flat, shape=(12,)).copy()
    assert got.dtype == np.int64 and np.array_equal(got.reshape(3, 4), np.asarray(rows, np.int64))
    t = model.create_tensor(flat, 3, 4)
    assert t and t.contents.type == 7 and t.contents.ndim == 2 and list(t.contents.dims[:2]) == [3, 4]
    assert t.contents.data == C.addressof(flat.contents) and t.contents.owns_data == 0      # wraps, does not copy/own
    api = C.c_void_p.in_dll(model, "g_ort")
    assert api.value
    marrs, mptrs = _ragged_rows([[1, 1, 1, 0], [1, 1, 0, 0], [1, 1, 1, 1]])
    tok = _lib.TokenizedInputs(ptrs, ptrs, mptrs, 3, 4)
    a, b = C.POINTER(_lib.OrtValue)(), C.POINTER(_lib.OrtValue)()
    assert model.prepare_input_tensors(C.byref(tok), C.byref(a), C.byref(b)) == 0
    assert list(a.contents.dims[:2]) == [3, 4] and a.contents.owns_data == 1 and b.contents.owns_data == 1
    ids = np.ctypeslib.as_array(C.cast(a.contents.data, C.POINTER(C.c_int64)), shape=(3, 4))
    mk = np.ctypeslib.as_array(C.cast(b.contents.data, C.POINTER(C.c_int64)), shape=(3, 4))
    assert np.array_equal(ids, np.asarray(rows)) and mk.sum() == 9
    assert model.prepare_input_tensors(None, C.byref(a), C.byref(b)) == -1


_POST = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %(root)r)
from gliclass.c_amd import _lib
m = _lib.model(); m.initialize_ort_api()
api = C.c_void_p.in_dll(m, "g_ort")
logits = np.array(%(logits)r, np.float32)
B, Cn = logits.shape
dims = (C.c_int64 * 2)(B, Cn)
class OrtApi(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("CreateEnv","ReleaseEnv","ReleaseSession","ReleaseValue","ReleaseStatus","GetErrorMessage",
                "CreateCpuMemoryInfo","ReleaseMemoryInfo","CreateTensorWithDataAsOrtValue")]
vt = C.cast(api, C.POINTER(OrtApi)).contents
mk = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p))(vt.CreateTensorWithDataAsOrtValue)
val = C.c_void_p()
assert mk(None, logits.ctypes.data, logits.nbytes, dims, 2, 1, C.byref(val)) is None
def strs(xs): return (C.c_char_p * len(xs))(*[x.encode() for x in xs])
labels = %(labels)r
lab_arrs = [strs(l) for l in labels]
lab = (C.POINTER(C.c_char_p) * len(labels))(*[C.cast(a, C.POINTER(C.c_char_p)) for a in lab_arrs])
nl = (C.c_size_t * len(labels))(*[len(l) for l in labels])
texts = strs(%(texts)r)
m.process_output_tensor.argtypes = [C.c_void_p, C.c_void_p, C.c_bool, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_size_t, C.c_void_p, C.c_char_p]
m.process_output_tensor(val, api, %(same)r, lab, nl, %(nls)d, C.c_float(%(thr)r), B, texts, %(kind)r)
'''


def _run_post(logits, labels, texts, same, kind, thr=0.5):
    code = _POST % dict(root=ROOT, logits=logits, labels=labels, texts=texts, same=same, nls=len(labels[0]), thr=thr, kind=kind.encode())
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout


def _sig(x):
    return float(np.float32(1.0) / (np.float32(1.0) + np.exp(-np.float32(x), dtype=np.float32)))


def test_process_output_tensor_formats_and_rules(libs):
    """Observable behaviour of /root/reference/src/postprocessor.c:88-150."""
    logits = [[2.0, -1.0, 0.0, 0.5], [-3.0, -2.0, -4.0, -5.0]]
    out = _run_post(logits, [["alpha", "beta", "gamma", "delta"]], ["first text", "second"], True, "multi-label")
    exp = ("Text_0: first text:\n  Text_0 Label: alpha, Score: %.6f\n  Text_0 Label: delta, Score: %.6f\n\nText_1: second:\n\n"
           % (_sig(2.0), _sig(0.5)))            # 0.0 -> prob 0.5 is NOT > 0.5 (strict compare, :95)
    assert out == exp
    out = _run_post(logits, [["alpha", "beta", "gamma", "delta"]], ["first text", "second"], True, "single-label")
    assert out == "Text_0: first text:\n  Text_0 Label: alpha, Score: %.6f\n\nText_1: second:\n  Text_1 Label: beta, Score: %.6f\n\n" % (_sig(2.0), _sig(-2.0))
    # per-text labels: slot j >= num_labels[i] prints [Unknown] (:101-111)
    out = _run_post([[3.0, 3.0], [3.0, -3.0]], [["a"], ["x", "y"]], ["t0", "t1"], False, "multi-label")
    assert out == ("Text_0: t0:\n  Text_0 Label: a, Score: %.6f\n  Text_0 Label: [Unknown], Score: %.6f\n\nText_1: t1:\n  Text_1 Label: x, Score: %.6f\n\n"
                   % (_sig(3.0), _sig(3.0), _sig(3.0)))
    assert _run_post(logits, [["a", "b", "c", "d"]], ["p", "q"], True, "ranking") == "This type of classification is not supported\n"


def test_sigmoid_matches_reference_formula(libs):
    _lib, hip, model = libs
    for x in (-30.0, -1.25, 0.0, 0.75, 12.0):
        assert abs(model.sigmoid(x) - 1.0 / (1.0 + np.exp(-x))) < 1e-7


def test_synthetic_inputs_layout():
    from gliclass.c_amd import synth
    from gliclass.c_amd.config import CONFIGS
    cfg = CONFIGS["base"]
    ids, mask, counts = synth.make_inputs(cfg, 4, 64, 8)
    assert (counts == 8).all() and mask.all() and (ids[:, 0] == cfg.cls_id).all() and (ids[:, -1] == cfg.sep_id).all()
    assert (ids[:, 1] == cfg.class_token_index).all() and (ids[:, 25] == cfg.text_token_index).all()
    ids, mask, counts = synth.make_inputs(cfg, 6, 200, 3, ragged=True)
    lens = mask.sum(1)
    assert lens.max() == 200 and lens.min() >= 100 and (ids[mask == 0] == 0).all()
    assert all(mask[b, :lens[b]].all() for b in range(6))        # prefix masks, like the tokenizer's padding


def _build_example(tmp_path):
    exe = str(tmp_path / "run_pretokenized")
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "run_pretokenized.c"), "-L" + os.path.join(ROOT, "gliclass", "c_amd"),
                           "-lgliclass_model", "-lgliclass_hip", "-Wl,-rpath," + os.path.join(ROOT, "gliclass", "c_amd"), "-fopenmp", "-o", exe])
    return exe


def test_pure_c_driver_builds_against_the_headers(libs, tmp_path):
    """examples/run_pretokenized.c is the reference's main.c call sequence in plain C11 (defines g_ort itself,
    /root/reference/main.c:33): it must compile warning-free against include/ and link against the two libraries."""
    _lib, hip, model = libs
    exe = _build_example(tmp_path)
    tok = tmp_path / "tok.txt"
    tok.write_text("1 513 7 8 514 20 21 2\n")
    r = subprocess.run([exe, "synthetic:tiny", str(tok), "multi-label", "a"], capture_output=True, text=True)
    if hip.glc_device_count() == 0:
        assert r.returncode != 0 and "no MI355X/HIP device" in r.stderr         # loud failure, no CPU path
    else:
        assert r.returncode == 0 and "Text_0" in r.stdout


def test_length_bucket_planner():
    """glc_plan_length_buckets (host only): groups are contiguous in the length-sorted order and cover every row once; the
    wave-quantised cost model keeps a 64-row batch of the base model in one piece (measured: splitting it is slower), splits a
    node-sized batch, and never splits equal lengths or reference-sized batches."""
    import ctypes as C
    import numpy as np
    from gliclass.c_amd import _lib
    L = _lib.hip()

    def plan(lengths, g, hidden=768):
        B = len(lengths)
        arr = (C.c_int * B)(*lengths)
        order, cuts, n = (C.c_int * B)(), (C.c_int * (B + 1))(), C.c_int(0)
        assert L.glc_plan_length_buckets(arr, B, g, hidden, order, cuts, C.byref(n)) == 0
        return list(order), list(cuts[: n.value + 1])

    def cost(lengths, order, cuts):
        rup = lambda x: (max(x, 1) + 63) // 64 * 64
        waves = lambda rows: -(-(-(-rows // 256) * 3) // 256)               # base model on 256 CUs: 3 tiles per 256 rows at N = 768
        return sum(waves((cuts[i + 1] - cuts[i]) * rup(lengths[order[cuts[i]]])) * 256 * 256 // 3 + 1024 for i in range(len(cuts) - 1))

    rng = np.random.default_rng(3)
    lengths = [int(x) for x in rng.integers(512, 1025, size=64)]
    order, cuts = plan(lengths, 4)
    assert sorted(order) == list(range(64)) and cuts[0] == 0 and cuts[-1] == 64 and cuts == sorted(cuts)
    assert all(lengths[order[i]] >= lengths[order[i + 1]] for i in range(63))
    assert len(cuts) == 2                                                             # c3-sized ragged batch: one piece
    big = [int(x) for x in rng.integers(512, 1025, size=256)]                          # a node-sized batch pays for groups
    ob, cb = plan(big, 4)
    assert 2 <= len(cb) - 1 <= 4 and cost(big, ob, cb) < 0.95 * cost(big, ob, [0, 256])
    assert cost(big, ob, cb) <= min(cost(big, ob, [0, j, 256]) for j in range(1, 256))  # at least as good as the best 2-split
    assert len(plan([1024, 900, 800, 700, 650, 600, 550, 520], 4)[1]) == 2
    assert len(plan([777] * 40, 4)[1]) == 2
    assert len(plan(big, 1)[1]) == 2


def _hf_config_json(cfg, **over):
    """config.json in the layout of a GLiClass checkpoint: backbone config under `encoder_config`, head fields on top."""
    if cfg.backbone == 0:
        enc = {"model_type": "deberta-v2", "hidden_size": cfg.hidden, "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads,
               "intermediate_size": cfg.inter, "vocab_size": cfg.vocab - 2, "relative_attention": True, "position_buckets": cfg.pos_buckets,
               "max_relative_positions": -1, "max_position_embeddings": cfg.max_rel_pos, "norm_rel_ebd": "layer_norm", "share_att_key": True,
               "pos_att_type": ["p2c", "c2p"], "position_biased_input": False, "type_vocab_size": 0, "layer_norm_eps": cfg.ln_eps, "pad_token_id": 0}
    else:
        enc = {"model_type": "qwen2", "hidden_size": cfg.hidden, "num_hidden_layers": cfg.layers, "num_attention_heads": cfg.heads,
               "num_key_value_heads": cfg.kv_heads, "intermediate_size": cfg.inter, "vocab_size": cfg.vocab - 2, "rms_norm_eps": cfg.ln_eps,
               "rope_theta": cfg.rope_theta}
    top = {"model_type": "GLiClass", "architecture_type": "uni-encoder", "encoder_config": enc, "class_token_index": cfg.class_token_index,
           "text_token_index": cfg.text_token_index, "pooling_strategy": {0: "first", 1: "avg", 2: "last"}[cfg.pooling],
           "scorer_type": {0: "simple", 1: "weighted-dot", 2: "mlp"}[cfg.scorer],
           "embed_class_token": True, "normalize_features": False, "use_lstm": False, "vocab_size": cfg.vocab}
    top.update(over)
    return top


def test_native_hf_checkpoint_import(libs, tmp_path):
    """SURVEY.md §8f-2 natively: config.json + model.safetensors (F32 / F16 / BF16, GLiClass-style prefixes) -> glc_weights_load;
    the tensors and the derived configuration equal the python importer's; unsupported configurations are refused."""
    import json
    torch = pytest.importorskip("torch")
    st = pytest.importorskip("safetensors.torch")
    _lib, hip, model = libs
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    import dataclasses
    from gliclass.c_amd.config import SCORER_NAMES
    for cname, prefix, scorer in (("tiny", "encoder_model.model.", "simple"), ("dec-tiny", "decoder_model.model.", "simple"),
                                  ("tiny", "encoder_model.model.", "weighted-dot"), ("dec-tiny", "decoder_model.model.", "mlp")):
        cfg = dataclasses.replace(CONFIGS[cname], scorer=SCORER_NAMES[scorer])
        ref = weights.make_weights(cfg, 11)
        names = [s[0] for s in weights.tensor_specs(cfg)]
        for dt in (torch.float32, torch.float16, torch.bfloat16) if scorer == "simple" else (torch.float32,):
            d = tmp_path / f"{cname}_{scorer}_{str(dt).split('.')[-1]}"
            d.mkdir()
            sd = {(prefix if "projector" not in n and not n.startswith("scorer.") else "") + n: torch.from_numpy(ref[n]).to(dt) for n in names}
            sd["some.unrelated.buffer"] = torch.zeros(3, dtype=torch.int64)
            st.save_file(sd, str(d / "model.safetensors"), metadata={"format": "pt"})
            (d / "config.json").write_text(json.dumps(_hf_config_json(cfg)))
            for src in (str(d), str(d / "model.safetensors")):
                W = _lib.Weights()
                assert model.glc_weights_load(src.encode(), C.byref(W)) == 0, src
                for f in ("vocab", "hidden", "layers", "heads", "head_dim", "inter", "pos_buckets", "max_rel_pos", "pad_id", "cls_id", "sep_id",
                          "class_token_index", "text_token_index", "pooling", "scorer", "embed_class_token", "normalize_features", "backbone",
                          "kv_heads", "causal"):
                    if cfg.backbone == 1 and f in ("pos_buckets", "max_rel_pos"):
                        continue
                    assert getattr(W.cfg, f) == getattr(cfg, f), (cname, f)
                assert abs(W.cfg.ln_eps - cfg.ln_eps) < 1e-12 and W.n_tensors == len(names)
                for i, n in enumerate(names):
                    got = np.ctypeslib.as_array(W.tensors[i], shape=(ref[n].size,))
                    want = torch.from_numpy(ref[n]).to(dt).float().numpy().ravel()
                    assert np.array_equal(got, want), (n, dt)
                model.glc_weights_free(C.byref(W))
    # refusals (message on stderr, -1)
    cfg = CONFIGS["tiny"]
    d = tmp_path / "tiny_simple_float32"
    for over in ({"scorer_type": "hopfield"}, {"scorer_type": "mlp"}, {"architecture_type": "bi-encoder"}, {"use_lstm": True}, {"pooling_strategy": "max"}):
        (d / "config.json").write_text(json.dumps(_hf_config_json(cfg, **over)))
        W = _lib.Weights()
        assert model.glc_weights_load(str(d).encode(), C.byref(W)) != 0, over
    bad = _hf_config_json(cfg)
    bad["encoder_config"]["share_att_key"] = False
    (d / "config.json").write_text(json.dumps(bad))
    W = _lib.Weights()
    assert model.glc_weights_load(str(d).encode(), C.byref(W)) != 0
    (d / "config.json").write_text(json.dumps(_hf_config_json(cfg)))
    os.remove(d / "model.safetensors")
    assert model.glc_weights_load(str(d).encode(), C.byref(W)) != 0


def test_effective_cpus_respects_affinity_and_env(monkeypatch):
    """hostinfo.effective_cpus (python) and glc_host_cpus (C, host/glc_cpus.h) size thread teams to what the job may really use."""
    from gliclass.c_amd import hostinfo
    n = hostinfo.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
    monkeypatch.setenv("OMP_NUM_THREADS", "1")
    assert hostinfo.effective_cpus() == 1
    monkeypatch.setenv("OMP_NUM_THREADS", "100000")
    assert hostinfo.effective_cpus() == n or hostinfo.effective_cpus() <= (os.cpu_count() or 1)


def test_gelu_logit_fit_of_the_gemm_epilogue_is_fp32_exact():
    """The 256-tile GEMM epilogues of the fp32 mode compute GELU as x / (1 + 2^(x P(x^2))) (csrc/glc_common.h glc_gelu2_f32; HF
    ACT2FN["gelu"] = 0.5 x (1 + erf(x / sqrt 2)), modeling_deberta_v2.py:393).  The constants are read from the header and the formula is
    evaluated here in float32, operation by operation: it must stay within one fp32 ulp-class distance (7e-7) of the erf form over the
    whole range, including the clamp at x^2 = 36 and the far tails."""
    from scipy.special import erfc
    src = open(os.path.join(ROOT, "gliclass", "c_amd", "csrc", "glc_common.h")).read()
    body = src[src.index("f32x2 glc_gelu2_f32("):]
    body = body[:body.index("return x * r;")]
    c = [float(m) for m in re.findall(r"\(?(-?\d\.\d+e[+-]\d+)f\)?", body)]
    assert len(c) == 7 and "36.0f" in body
    f = np.float32
    xs = np.concatenate([np.linspace(-14, 14, 700001), np.linspace(-60, 60, 1201), [0.0, -0.0, 5.9999, 6.0, 6.0001, -6.0]]).astype(f)
    t = np.minimum(xs * xs, f(36.0))
    p = (t * f(c[0]) + f(c[1])).astype(f)
    for k in c[2:]:
        p = (p * t + f(k)).astype(f)
    with np.errstate(over="ignore"):
        g = (xs * (f(1.0) / (f(1.0) + np.exp2((xs * p).astype(f)).astype(f))).astype(f)).astype(f)
    x64 = xs.astype(np.float64)
    ref = x64 * 0.5 * erfc(-x64 / np.sqrt(2.0))
    assert np.isfinite(g).all()
    assert np.abs(g - ref).max() <= 7e-7, np.abs(g - ref).max()
    assert np.abs(g - ref)[np.abs(x64) <= 1.0].max() <= 1.5e-7
