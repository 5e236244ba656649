"""Generates tests/golden/reference_probe.json: the reference's OWN fixed probe as a named case.

  text / labels : /root/reference/ONNX_CONVERTING/convert_to_onnx.py:57-58 and test_onnx.py:64-65
  tolerance     : atol 1e-3 on the logits, /root/reference/ONNX_CONVERTING/test_onnx.py:30
  golden logits : `original_logits` of the model repo's onnx/config.json on the HF hub (convert_to_onnx.py:15 rounds them to 5 dp) —
                  NOT in the reference repo and not reachable offline => the case is recorded UNPINNED (original_logits: null).

What CAN be fixed offline and is stored: the two prompt layouts the reference builds for it (src/preprocessor.c:84-108: labels
lower-cased, "<<LABEL>>" before each, "<<SEP>>", text before or after) and their ids under the stand-in DeBERTa-v3-structured
tokenizer.json of tests/golden/ as produced by the Rust `tokenizers` library (what tokenizers-cpp wraps).  Run in the build
container (needs the `tokenizers` wheel):   python oracle/gen_reference_probe.py
"""
import gzip
import json
import os

from tokenizers import Tokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEXT = "ONNX is an open-source format designed to enable the interoperability of AI models across various frameworks and tools."
LABELS = ["format", "model", "tool", "cat"]


def main():
    tk = Tokenizer.from_str(gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer.json.gz")).read().decode("utf-8"))
    head = "".join("<<LABEL>>" + l.lower() for l in LABELS) + "<<SEP>>"
    prompts = {"prompt_first_true": head + TEXT, "prompt_first_false": TEXT + head}
    out = {
        "source": "/root/reference/ONNX_CONVERTING/convert_to_onnx.py:57-58, test_onnx.py:64-65",
        "text": TEXT, "labels": LABELS, "classification_type": "multi-label",
        "tolerance_atol": 1e-3, "tolerance_source": "/root/reference/ONNX_CONVERTING/test_onnx.py:30",
        "original_logits": None,
        "original_logits_note": "stored in onnx/config.json of the model repo on the HF hub (convert_to_onnx.py:15); unreachable offline: UNPINNED",
        "prompts": prompts,
        "ids_standin_tokenizer": {k: tk.encode(v, add_special_tokens=True).ids for k, v in prompts.items()},
        "tokenizer": "tests/golden/tokenizer.json.gz (stand-in with DeBERTa-v3 structure; the real tokenizer.json of a GLiClass model is not on disk)",
    }
    with open(os.path.join(ROOT, "tests", "golden", "reference_probe.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: len(v) for k, v in out["ids_standin_tokenizer"].items()})


if __name__ == "__main__":
    main()
