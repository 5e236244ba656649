"""Drop-in proof against the reference's OWN host sources: oracle/_ref/ref_main is /root/reference/main.c +
src/{parallel_processor,postprocessor,preprocessor,tokenizer}.c, compiled unchanged against this repo's include/ and linked against
libgliclass_model.so where the reference links src/model.c + ONNXRuntime, src/read_data.c + cJSON and tokenizers-cpp
(recipe: oracle/Makefile, target `ref`; built where /root/reference exists, the binary travels to the GPU box).

CPU: the reference launcher links, parses the JSON through this repo's read_data, builds prompts and a tokenizer through its own
preprocessor.c / tokenizer.c over this repo's tokenizers_c API, and fails loudly at create_ort_session for want of a GPU.
GPU: the whole reference launcher runs over the HIP engine (its OpenMP team calls run_inference concurrently, main.c:141-149) and
prints the same score lines as this repo's launcher and as the CPU oracle."""
import dataclasses
import gzip
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "ref_main")
OUR_MAIN = os.path.join(ROOT, "gliclass", "c_amd", "gliclass_main")

TEXTS = ["One day I will see the world!", "ONNX is an open-source format designed to enable the interoperability of AI models.",
         "The match ended 3:2 after extra time.", "Stocks fell sharply on Monday.", "She published a paper on protein folding.",
         "Vote on the new budget is due.", "Short.", "A " + "very " * 40 + "long sentence about travelling the world.",
         "The reference batches eight texts at a time.", "This is the ninth text, so the second batch is short.", "Eleven is prime."]
LABELS = ["travel", "dreams", "sport", "science", "politics"]


@pytest.fixture(scope="module")
def ref_main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gliclass", "c_amd"), "-j4", "all"], stdout=subprocess.DEVNULL)
    if os.path.isfile("/root/reference/main.c"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    if not os.path.isfile(REF_MAIN):
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/ref_main is present")
    return REF_MAIN


@pytest.fixture()
def rundir(tmp_path):
    """cwd laid out as the reference expects (include/paths.h): tokenizer/tokenizer.json and the model file."""
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    cfg = dataclasses.replace(CONFIGS["tiny"], name="tiny-tok", vocab=6003, class_token_index=6001, text_token_index=6002)
    w = weights.make_weights(cfg, 3)
    (tmp_path / "tokenizer").mkdir()
    (tmp_path / "tokenizer" / "tokenizer.json").write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer.json.gz")).read())
    (tmp_path / "model").mkdir()
    weights.write_blob(str(tmp_path / "model" / "model.glcw"), cfg, w)
    (tmp_path / "data.json").write_text(json.dumps({"texts": TEXTS, "labels": [LABELS], "same_labels": True, "classification_type": "multi-label"}))
    return tmp_path, cfg, w


def test_reference_launcher_links_and_reaches_the_session_boundary(ref_main, rundir):
    from gliclass.c_amd import _lib
    d, _, _ = rundir
    r = subprocess.run([ref_main], capture_output=True, text=True, cwd=d)
    assert r.returncode == 1 and r.stdout.startswith("Usage:")                        # /root/reference/main.c:54-61
    if _lib.hip().glc_device_count() > 0:
        pytest.skip("GPU present: the full run is the -m gpu test below")
    r = subprocess.run([ref_main, "data.json", "true"], capture_output=True, text=True, cwd=d)
    for stage in ("parse_json", "create_tokenizer", "initialize_ort_api", "initialize_ort_environment"):
        assert f"DONE: {stage};" in r.stdout, r.stdout
    assert "no MI355X/HIP device" in r.stderr and "Failed to create session" in r.stderr
    assert r.returncode == 255                                                        # main.c:93-97 returns -1


def _score_lines(stdout):
    return sorted(l for l in stdout.splitlines() if re.match(r"  Text_\d+ Label: .+, Score: [0-9.]+$", l))


@pytest.mark.gpu
def test_reference_launcher_runs_over_the_hip_engine(ref_main, rundir):
    import oracle_c
    from gliclass.c_amd.tokenizer import Tokenizer
    d, cfg, w = rundir
    env = {k: v for k, v in os.environ.items() if not k.startswith("GLICLASS_")}     # product defaults: f32 mode, threshold 0.5, batches of 8
    r = subprocess.run([ref_main, "data.json", "true"], capture_output=True, text=True, cwd=d, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    for stage in ("parse_json", "create_tokenizer", "initialize_ort_api", "initialize_ort_environment", "create_ort_session"):
        assert f"DONE: {stage};" in r.stdout
    assert "Execution time:" in r.stdout
    ours = subprocess.run([OUR_MAIN, "data.json", "true", "tokenizer/tokenizer.json", "model/model.glcw"], capture_output=True, text=True,
                          cwd=d, env=dict(env, GLICLASS_PIPELINE="0"))
    assert ours.returncode == 0, ours.stderr[-2000:]
    got = _score_lines(r.stdout)
    assert got and got == _score_lines(ours.stdout)               # the reference's printf path and this repo's: same bytes
    # and against the oracle: same prompts (src/preprocessor.c:84-108, prompt first), python twin of the native tokenizer, batches of 8
    tok = Tokenizer((d / "tokenizer" / "tokenizer.json").read_text())
    prompts = ["".join("<<LABEL>>" + l.lower() for l in LABELS) + "<<SEP>>" + t for t in TEXTS]
    want = []
    for lo in range(0, len(TEXTS), 8):
        ids, mask = tok.tokenize_inputs(prompts[lo:lo + 8], 2048)
        lg = oracle_c.forward(cfg, w, np.array(ids, np.int64), np.array(mask, np.int64))
        p = 1.0 / (1.0 + np.exp(-lg.astype(np.float64)))
        want += [float(x) for x in p.ravel() if x > 0.5]                              # THRESHOLD 0.5, strict > (postprocessor.c:95)
    scores = sorted(float(l.rsplit("Score: ", 1)[1]) for l in got)
    assert len(scores) == len(want)
    assert np.abs(np.array(scores) - np.array(sorted(want))).max() <= 2e-5            # %.6f printing + fp32
