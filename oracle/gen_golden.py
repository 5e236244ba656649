"""ORACLE tooling — regenerates tests/golden/*.npz from HuggingFace DebertaV2Model / Qwen2Model (fp32, CPU).

Run in the build container:  python oracle/gen_golden.py [case-name prefix]      (e.g. `dec_` for the decoder cases only)
The fixtures are *data* (inputs + expected outputs); weights are not stored — they are
reproduced from (config name, seed) by gliclass.c_amd.weights.make_weights on every side.

Cases mirror SURVEY.md §8c: tiny/mini configs at S in {16,128,200,600,1024} with ragged masks and
1-5 <<LABEL>> tokens per row, plus the known-answer for BASELINE config c1
(gliclass-small shape, B=1, S=128, C=4).  The reference's own probe sentence
(/root/reference/ONNX_CONVERTING/convert_to_onnx.py:57-58) cannot be tokenised offline
(no tokenizer.json) and its golden logits are on the HF hub -> recorded as UNPINNED in DESIGN.md.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import transformers  # noqa: E402
from transformers.models.deberta_v2.modeling_deberta_v2 import make_log_bucket_position  # noqa: E402

from gliclass.c_amd.config import CONFIGS  # noqa: E402
from gliclass.c_amd import weights, synth  # noqa: E402
import hf_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 42

# (case name, config, B, S, C, ragged, labels_per_row)
CASES = [
    ("tiny_b1_s16", "tiny", 1, 16, 1, False, None),
    ("tiny_b3_s200", "tiny", 3, 200, 4, True, [4, 2, 3]),
    ("tiny_b2_s600", "tiny", 2, 600, 5, True, [5, 1]),
    ("tiny_b2_s1024", "tiny", 2, 1024, 3, True, None),
    ("mini_b4_s128", "mini", 4, 128, 4, True, [4, 4, 1, 3]),
    ("mini_b2_s333", "mini", 2, 333, 2, True, None),
    ("small_c1_b1_s128", "small", 1, 128, 4, False, None),   # BASELINE.json configs[0] shape
    # decoder-style backbone (BASELINE.json configs[4] arithmetic: RoPE, grouped-query causal attention, SwiGLU, RMSNorm)
    ("dec_tiny_b3_s200", "dec-tiny", 3, 200, 4, True, [4, 2, 3]),
    ("dec_tiny_b2_s700", "dec-tiny", 2, 700, 5, True, [5, 1]),
    ("dec_mini_b2_s333", "dec-mini", 2, 333, 3, True, None),
]


def sample_positions(S):
    pos = sorted(set([0, 1, 2, 4, 7, 13, S // 3, S // 2, S - 2, S - 1]) & set(range(S)))
    return np.asarray(pos, np.int64)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    os.makedirs(OUT, exist_ok=True)
    meta = dict(transformers=transformers.__version__, torch=torch.__version__, weight_seed=WEIGHT_SEED)
    # 1) relative-position tables, straight from HF's jit-scripted float32 function
    tabs = {}
    for S in (16, 128, 200, 333, 600, 1024, 2048, 4096):
        rel = torch.arange(-(S - 1), S, dtype=torch.long)
        b = make_log_bucket_position(rel, 256, 512).to(torch.long)
        tabs[f"S{S}"] = torch.clamp(b + 256, 0, 511).numpy().astype(np.int16)
    if not only:
        np.savez_compressed(os.path.join(OUT, "delta_tables.npz"), **tabs)

    models = {}
    for name, cname, B, S, C, ragged, lpr in CASES:
        if not name.startswith(only):
            continue
        cfg = CONFIGS[cname]
        if cname not in models:
            w = weights.make_weights(cfg, WEIGHT_SEED)
            models[cname] = (w, hf_ref.build_hf_model(cfg, w))
        w, model = models[cname]
        ids, mask, counts = synth.make_inputs(cfg, B, S, C, seed=1234 + S, ragged=ragged, labels_per_row=lpr)
        logits, hs = hf_ref.forward(cfg, w, ids, mask, model=model, want_hidden=True)
        hs = np.stack(hs)                                # [L+1, B, S, H]
        pos = sample_positions(S)
        rec = dict(
            config=np.array(cname), B=B, S=S, ids=ids.astype(np.int32), mask=mask.astype(np.int8),
            counts=counts.astype(np.int32), logits=logits.astype(np.float32),
            probs=(1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.float32),
            sample_pos=pos, hidden_samples=hs[:, :, pos, :].astype(np.float32),
            hidden_abs_sum=np.abs(hs * mask[None, :, :, None]).sum(axis=(2, 3)).astype(np.float64),
            meta=np.array(str(meta)),
        )
        if cname != "small":
            rec["hidden_samples"] = rec["hidden_samples"][..., : min(cfg.hidden, 128)]
        else:
            rec["hidden_samples"] = rec["hidden_samples"][:, :, :4, :64]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, "logits", np.round(logits[0], 4), "bytes", os.path.getsize(os.path.join(OUT, name + ".npz")))


if __name__ == "__main__":
    main()
