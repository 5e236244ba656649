"""The launcher examples/gliclass_main.c (drop-in for /root/reference/main.c): CPU = front half (JSON -> prompts -> native
tokenizer) and the loud failure without a GPU; GPU = whole pipeline text -> printed scores against the CPU oracle fed with
ids from the SAME prompts tokenized by the python path."""
import dataclasses
import gzip
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "gliclass", "c_amd", "gliclass_main")

TEXTS = ["One day I will see the world!", "ONNX is an open-source format designed to enable the interoperability of AI models.",
         "The match ended 3:2 after extra time.", "Der schnelle braune Fuchs springt über den faulen Hund.", "Stocks fell sharply on Monday.",
         "", "She published a paper on protein folding.", "El veloz murciélago hindú comía feliz cardillo y kiwi.", "Vote on the new budget is due.",
         "Short.", "A " + "very " * 60 + "long sentence about travelling the world and dreaming of science."]
LABELS = ["Travel", "dreams", "SPORT", "science", "politics"]


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gliclass", "c_amd"), "-j4", "all"], stdout=subprocess.DEVNULL)
    d = tmp_path_factory.mktemp("cli")
    (d / "tok.json").write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer.json.gz")).read())
    return d


def _write_data(d, same, ctype, name="data.json"):
    labels = [LABELS] if same else [LABELS[: 1 + i % len(LABELS)] for i in range(len(TEXTS))]
    (d / name).write_text(json.dumps({"texts": TEXTS, "labels": labels, "same_labels": same, "classification_type": ctype}))
    return str(d / name), labels


def test_usage_and_front_half_without_gpu(workdir):
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("Usage:")                       # main.c:54-61
    data, _ = _write_data(workdir, True, "multi-label")
    r = subprocess.run([EXE, data, "maybe"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid value for bool argument" in r.stdout         # read_data.c:165
    (workdir / "notype.json").write_text(json.dumps({"texts": ["x"], "labels": [["a"]], "same_labels": True}))
    r = subprocess.run([EXE, str(workdir / "notype.json"), "true"], capture_output=True, text=True)
    assert r.returncode == 1 and "classification type is not provided" in r.stdout    # main.c:71-74
    r = subprocess.run([EXE, data, "true", str(workdir / "missing.json")], capture_output=True, text=True)
    assert r.returncode == 1 and "Cant open file" in r.stderr                          # tokenizer.c:149
    # prompt_first = auto: read from the model directory's config.json (run_GLiClass.sh:84-89)
    mdir = workdir / "model_dir"
    mdir.mkdir(exist_ok=True)
    (mdir / "config.json").write_text(json.dumps({"prompt_first": "yes"}))
    r = subprocess.run([EXE, data, "auto", str(workdir / "tok.json"), str(mdir)], capture_output=True, text=True)
    assert r.returncode == 1 and "Something wrong with model configuration file." in r.stderr
    (mdir / "config.json").write_text(json.dumps({"prompt_first": True, "encoder_config": {}}))
    r = subprocess.run([EXE, data, "auto", str(workdir / "tok.json"), str(mdir)], capture_output=True, text=True)
    assert "DONE: create_tokenizer;" in r.stdout and r.returncode != 0          # got past prompt_first; the (empty) checkpoint is refused
    from gliclass.c_amd import _lib
    if _lib.hip().glc_device_count() == 0:
        r = subprocess.run([EXE, data, "true", str(workdir / "tok.json"), "synthetic:tiny"], capture_output=True, text=True)
        assert r.returncode == 255 and "DONE: create_tokenizer;" in r.stdout and "no MI355X/HIP device" in r.stderr


def _prompts(labels_for, prompt_first):
    out = []
    for i, t in enumerate(TEXTS):                                                    # src/preprocessor.c:84-108
        p = "".join("<<LABEL>>" + l.lower() for l in labels_for(i)) + "<<SEP>>"
        out.append(p + t if prompt_first else t + p)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("same,prompt_first,ctype,pipeline", [(True, True, "multi-label", "1"), (False, False, "multi-label", "0"),
                                                                (True, False, "single-label", "1"), (False, True, "multi-label", "1")])
def test_launcher_end_to_end_vs_oracle(workdir, same, prompt_first, ctype, pipeline):
    import oracle_c
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.tokenizer import Tokenizer
    cfg = dataclasses.replace(CONFIGS["tiny"], name="tiny-tok", vocab=6003, class_token_index=6001, text_token_index=6002)
    w = weights.make_weights(cfg, 3)
    blob = str(workdir / "tiny_tok.glcw")
    weights.write_blob(blob, cfg, w)
    data, labels = _write_data(workdir, same, ctype, f"data_{int(same)}{int(prompt_first)}.json")
    labels_for = (lambda i: labels[0]) if same else (lambda i: labels[i])
    env = dict(os.environ, GLICLASS_DTYPE="f32", GLICLASS_THRESHOLD="0.0", GLICLASS_PIPELINE=pipeline)
    r = subprocess.run([EXE, data, "true" if prompt_first else "false", str(workdir / "tok.json"), blob], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    for stage in ("parse_json", "create_tokenizer", "initialize_ort_api", "initialize_ort_environment", "create_ort_session"):
        assert f"DONE: {stage};" in r.stdout
    assert "Execution time:" in r.stdout
    # expected: the same prompts, tokenized by the native tokenizer (pinned to HF `tokenizers` in test_tokenizer.py), through the oracle
    tok = Tokenizer((workdir / "tok.json").read_text())
    prompts = _prompts(labels_for, prompt_first)
    want = {}
    for lo in range(0, len(TEXTS), 8):                                               # BATCH_SIZE chunks (parallel_processor.c:28-30)
        ids, mask = tok.tokenize_inputs(prompts[lo:lo + 8], 2048)
        ids, mask = np.array(ids, np.int64), np.array(mask, np.int64)
        assert ((ids == 6001).sum(1) == [len(labels_for(i)) for i in range(lo, min(lo + 8, len(TEXTS)))]).all()
        lg = oracle_c.forward(cfg, w, ids, mask)
        p = 1.0 / (1.0 + np.exp(-lg.astype(np.float64)))
        for b in range(ids.shape[0]):
            want[lo + b] = {labels_for(lo + b)[j]: p[b, j] for j in range(len(labels_for(lo + b)))}
    got, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.match(r"Text_(\d+): (.*):$", line)
        m2 = re.match(r"  Text_(\d+) Label: (.+), Score: ([0-9.]+)$", line)
        if m2 and m2.group(2) != "[Unknown]":      # padded class slots of rows with fewer labels print as [Unknown] (postprocessor.c:107-109)
            got.setdefault(("local", int(m2.group(1))), []).append((m2.group(2), float(m2.group(3))))
    # printed indices are batch-local (postprocessor.c:90): text i of batch k prints as Text_i; collect per (batch, local) by order
    n_lines = sum(len(v) for v in got.values())
    if ctype == "multi-label":
        assert n_lines == sum(len(v) for v in want.values())                         # threshold 0.0: every label printed
        flat_got = sorted(s for v in got.values() for _, s in v)
        flat_want = sorted(s for v in want.values() for s in v.values())
        assert np.abs(np.array(flat_got) - np.array(flat_want)).max() <= 2e-5        # %.6f + fp32
    else:
        assert n_lines == len(TEXTS)                                                 # one argmax line per text
        flat_got = sorted(s for v in got.values() for _, s in v)
        flat_want = sorted(max(v.values()) for v in want.values())
        assert np.abs(np.array(flat_got) - np.array(flat_want)).max() <= 2e-5


@pytest.mark.gpu
def test_pipelined_stages_print_the_same_blocks_in_batch_order(workdir):
    """parallel_classify (pre / inference / post pipelined per batch) against the reference's three phases
    (GLICLASS_PIPELINE=0): the same per-text output blocks; the pipelined run prints them in batch order."""
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    cfg = dataclasses.replace(CONFIGS["tiny"], name="tiny-tok", vocab=6003, class_token_index=6001, text_token_index=6002)
    blob = str(workdir / "tiny_tok_p.glcw")
    weights.write_blob(blob, cfg, weights.make_weights(cfg, 3))
    texts = [f"{i}: " + TEXTS[i % len(TEXTS)] + " extra words" * (i % 7) for i in range(53)]          # 14 batches of 4, last one short
    (workdir / "many.json").write_text(json.dumps({"texts": texts, "labels": [LABELS], "same_labels": True, "classification_type": "multi-label"}))

    def run(pipeline, threads):
        env = dict(os.environ, GLICLASS_DTYPE="f32", GLICLASS_THRESHOLD="0.3", GLICLASS_PIPELINE=pipeline, GLICLASS_BATCH_SIZE="4",
                   GLICLASS_PIPELINE_THREADS=threads)
        r = subprocess.run([EXE, str(workdir / "many.json"), "true", str(workdir / "tok.json"), blob], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        body = r.stdout.split("DONE: create_ort_session;\n\n", 1)[1].rsplit("Execution time:", 1)[0]
        return [blk for blk in body.split("\n\n") if blk.strip()]

    three_phase = run("0", "1")
    for threads in ("1", "3", "8"):
        piped = run("1", threads)
        assert sorted(piped) == sorted(three_phase) and len(piped) == len(texts)
        order = [int(re.match(r"Text_\d+: (\d+): ", blk).group(1)) for blk in piped]
        assert order == list(range(len(texts)))                                        # deterministic: batch order, then row order


@pytest.mark.gpu
def test_multi_engine_session_on_one_gpu(workdir):
    """The single-process multi-GPU path of the drop-in layer (GLICLASS_DEVICES lists the engines of a session; run_inference deals
    concurrent calls round-robin, every engine has its own coalescing queue) exercised with two engines on the ONE GPU of a test box:
    same printed blocks, same order as the single-engine session."""
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    cfg = dataclasses.replace(CONFIGS["tiny"], name="tiny-tok", vocab=6003, class_token_index=6001, text_token_index=6002)
    blob = str(workdir / "tiny_tok_m.glcw")
    weights.write_blob(blob, cfg, weights.make_weights(cfg, 3))
    texts = [f"{i}: " + TEXTS[i % len(TEXTS)] for i in range(41)]
    (workdir / "multi.json").write_text(json.dumps({"texts": texts, "labels": [LABELS], "same_labels": True, "classification_type": "multi-label"}))

    def run(devices, pipeline):
        env = dict(os.environ, GLICLASS_DTYPE="f32", GLICLASS_THRESHOLD="0.3", GLICLASS_BATCH_SIZE="4", GLICLASS_DEVICES=devices,
                   GLICLASS_PIPELINE=pipeline)
        r = subprocess.run([EXE, str(workdir / "multi.json"), "true", str(workdir / "tok.json"), blob], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert f"on {len(devices.split(','))} GPU(s)" in r.stdout
        body = r.stdout.split("DONE: create_ort_session;\n\n", 1)[1].rsplit("Execution time:", 1)[0]
        return [blk for blk in body.split("\n\n") if blk.strip()]

    one = run("0", "1")
    assert run("0,0", "1") == one                          # pipelined: ordered retirement -> identical output
    assert sorted(run("0,0,0", "0")) == sorted(one)        # three phases (parallel_inference: one host thread per engine)
    from gliclass.c_amd import _lib
    if _lib.hip().glc_device_count() >= 2:                 # two DISTINCT devices when the box has them
        assert run("0,1", "1") == one
        assert sorted(run("0,1", "0")) == sorted(one)


@pytest.mark.gpu
@pytest.mark.parametrize("prompt_first", [True, False])
def test_reference_probe_through_the_launcher(workdir, prompt_first):
    """The reference's own probe (convert_to_onnx.py:57-58; named fixture tests/golden/reference_probe.json) text-in -> printed
    scores, against the oracle on the fixture's ids.  Its hub-stored golden logits are unreachable, so this pins the PATH the
    probe takes (prompt, tokenizer, forward, sigmoid, printing) on a synthetic model, not the trained model's numbers."""
    import oracle_c
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    probe = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_probe.json")))
    cfg = dataclasses.replace(CONFIGS["tiny"], name="tiny-tok", vocab=6003, class_token_index=6001, text_token_index=6002)
    w = weights.make_weights(cfg, 3)
    blob = str(workdir / "tiny_tok_probe.glcw")
    weights.write_blob(blob, cfg, w)
    data = workdir / f"probe_{int(prompt_first)}.json"
    data.write_text(json.dumps({"texts": [probe["text"]], "labels": [probe["labels"]], "same_labels": True,
                                "classification_type": probe["classification_type"]}))
    env = dict({k: v for k, v in os.environ.items() if not k.startswith("GLICLASS_")}, GLICLASS_THRESHOLD="0.0")     # default mode
    r = subprocess.run([EXE, str(data), "true" if prompt_first else "false", str(workdir / "tok.json"), blob], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    ids = np.array([probe["ids_standin_tokenizer"]["prompt_first_true" if prompt_first else "prompt_first_false"]], np.int64)
    assert int((ids == 6001).sum()) == len(probe["labels"])
    lg = oracle_c.forward(cfg, w, ids, np.ones_like(ids))
    want = {l: 1.0 / (1.0 + np.exp(-float(lg[0, j]))) for j, l in enumerate(probe["labels"])}
    got = {m.group(1): float(m.group(2)) for m in re.finditer(r"  Text_0 Label: (\w+), Score: ([0-9.]+)", r.stdout)}
    assert set(got) == set(want)
    assert max(abs(got[k] - want[k]) for k in want) <= probe["tolerance_atol"] * 0.02       # 2e-5: %.6f printing + fp32, 50x inside the probe's atol


@pytest.mark.gpu
def test_decoder_backbone_text_in_with_the_bpe_tokenizer(workdir):
    """Config c5's path text-in: the decoder-style backbone behind the launcher with a byte-level BPE tokenizer.json (the tokenizer
    family of the reference's qwen / llama GLiClass models, Readme.md:91-94; stand-in file tests/golden/bpe_tokenizer.json.gz, ids
    pinned to the Rust library in tests/test_tokenizer_bpe.py).  Printed scores against the oracle on the same prompts."""
    import oracle_c
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd.tokenizer import Tokenizer
    (workdir / "bpe_tok.json").write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "bpe_tokenizer.json.gz")).read())
    cfg = dataclasses.replace(CONFIGS["dec-tiny"], name="dec-tiny-tok", vocab=6002, class_token_index=6000, text_token_index=6001)
    w = weights.make_weights(cfg, 3)
    blob = str(workdir / "dec_tiny_tok.glcw")
    weights.write_blob(blob, cfg, w)
    data, labels = _write_data(workdir, False, "multi-label", "data_dec.json")
    env = dict({k: v for k, v in os.environ.items() if not k.startswith("GLICLASS_")}, GLICLASS_THRESHOLD="0.0")      # default mode
    r = subprocess.run([EXE, data, "true", str(workdir / "bpe_tok.json"), blob], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    tok = Tokenizer((workdir / "bpe_tok.json").read_text())
    prompts = _prompts(lambda i: labels[i], True)
    want = []
    for lo in range(0, len(TEXTS), 8):
        ids, mask = tok.tokenize_inputs(prompts[lo:lo + 8], 2048)
        ids, mask = np.array(ids, np.int64), np.array(mask, np.int64)
        assert ((ids == 6000).sum(1) == [len(labels[i]) for i in range(lo, min(lo + 8, len(TEXTS)))]).all()
        lg = oracle_c.forward(cfg, w, ids, mask)
        p = 1.0 / (1.0 + np.exp(-lg.astype(np.float64)))
        for b in range(ids.shape[0]):
            want += [p[b, j] for j in range(len(labels[lo + b]))]
    got = [float(m.group(1)) for m in re.finditer(r"  Text_\d+ Label: (?!\[Unknown\]).+, Score: ([0-9.]+)", r.stdout)]
    assert len(got) == len(want)
    assert np.abs(np.array(sorted(got)) - np.array(sorted(want))).max() <= 2e-5
