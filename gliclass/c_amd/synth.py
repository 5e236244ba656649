"""Synthetic *tokenized* inputs.

No tokenizer.json exists offline, so benches/tests start where the reference's
`tokenize_inputs` ends (`/root/reference/src/tokenizer.c:19-91`): int ids + 0/1 mask, padded to
the longest row with id 0 / mask 0.  Row layout mirrors the prompt-first layout of
`/root/reference/src/preprocessor.c:84-108`:  [CLS] (<<LABEL>> w w)*C <<SEP>> text... [SEP] pad...
"""
import numpy as np

from .config import GLiClassConfig
from . import prng


def make_inputs(cfg: GLiClassConfig, B: int, S: int, C: int, seed: int = 1234, ragged: bool = False,
                labels_per_row=None):
    """Returns (ids int64 [B,S], mask int64 [B,S], num_labels [B])."""
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), np.int64)
    words = prng.randint(seed, "ids", B * S, 3, cfg.vocab - 2).reshape(B, S)
    lens = prng.randint(seed, "lens", B, max(S // 2, 1), S + 1) if ragged else np.full(B, S)
    if ragged:
        lens[0] = S                      # keep pad-to-longest semantics: one row defines S
    nl = np.asarray(labels_per_row if labels_per_row is not None else [C] * B, np.int64)
    for b in range(B):
        c = int(nl[b])
        prefix = 1 + 3 * c + 1
        n = int(max(lens[b], min(prefix + 2, S)))
        n = min(n, S)
        row = words[b].copy()
        row[0] = cfg.cls_id
        p = 1
        for _ in range(c):
            if p + 3 > S - 2:
                break
            row[p] = cfg.class_token_index
            p += 3
        if p < S - 1:
            row[p] = cfg.text_token_index
        row[n - 1] = cfg.sep_id
        ids[b, :n] = row[:n]
        mask[b, :n] = 1
    counts = (ids == cfg.class_token_index).sum(1)
    return ids, mask, counts
