"""CPU: pins the C oracle (oracle/gliclass_oracle.c) on the HF-generated golden fixtures."""
import glob
import os

import numpy as np
import pytest

import oracle_c

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*_b*_s*.npz")))


def test_delta_table_bit_exact(golden_dir):
    """The log-bucket table must match torch's float32 arithmetic bit for bit (SURVEY.md §7 hard part 2)."""
    tabs = np.load(os.path.join(golden_dir, "delta_tables.npz"))
    for key in tabs.files:
        S = int(key[1:])
        got = oracle_c.delta_table(S)
        assert np.array_equal(got, tabs[key].astype(np.int32)), key
    t = oracle_c.delta_table(1024)
    assert t[1023] == 256 and t[0] == 0 and t[-1] == 511          # clamp asymmetry for S > 512
    assert t[1023 + 127] == 383 and t[1023 - 127] == 129          # identity region edge


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_golden(case, golden_dir, weights_for):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    cfg, w = weights_for(str(g["config"]))
    ids, mask = g["ids"].astype(np.int64), g["mask"].astype(np.int64)
    logits, hidden = oracle_c.forward(cfg, w, ids, mask, want_hidden=True)
    assert logits.shape == g["logits"].shape
    np.testing.assert_allclose(logits, g["logits"], atol=1e-4, rtol=0)   # fp32 summation-order noise; bar is 1e-3
    probs = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    np.testing.assert_allclose(probs, g["probs"], atol=3e-5, rtol=0)
    pos = g["sample_pos"]
    hs = g["hidden_samples"]
    got = hidden[:, :, pos, :][..., : hs.shape[-1]][:, :, : hs.shape[2]]
    valid = mask[:, pos][:, : hs.shape[2]].astype(bool)
    np.testing.assert_allclose(got[:, valid], hs[:, valid], atol=2e-4, rtol=0)
    s = np.abs(hidden * mask[None, :, :, None]).sum(axis=(2, 3))
    np.testing.assert_allclose(s, g["hidden_abs_sum"], rtol=1e-5)


def test_oracle_edge_cases(weights_for):
    """Ragged C, a row without labels, S=1-token-after-prefix, all-padding tail."""
    from gliclass.c_amd import synth
    cfg, w = weights_for("tiny")
    ids, mask, counts = synth.make_inputs(cfg, 3, 40, 3, ragged=True, labels_per_row=[3, 0, 1])
    assert list(counts) == [3, 0, 1]
    logits = oracle_c.forward(cfg, w, ids, mask)
    assert logits.shape == (3, 3) and np.isfinite(logits).all()
    # rows are independent: running row 2 alone (trimmed to its own length) gives the same logits
    n = int(mask[2].sum())
    solo = oracle_c.forward(cfg, w, ids[2:3, :n], mask[2:3, :n])
    np.testing.assert_allclose(solo[0, :1], logits[2, :1], atol=2e-5)


def test_sigmoid_matches_reference_formula():
    lib = oracle_c.lib()
    for x in (-20.0, -1.5, 0.0, 0.3, 7.0):
        assert abs(lib.glo_sigmoid(x) - 1.0 / (1.0 + np.exp(-x))) < 1e-7


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("scorer", ["weighted-dot", "mlp"])
def test_oracle_scorer_variants_vs_torch_modules(scorer, normalize):
    """The head's other scorers ('must be config-switchable'): the C oracle against the same modules written
    with torch.nn, on the oracle's own final hidden states.  The module structure is the upstream `gliclass` package's as restated in
    include/gliclass_hip.h (not on disk here: parity with upstream stays unpinned); this pins the oracle's arithmetic to torch's."""
    import dataclasses
    import torch
    from gliclass.c_amd.config import CONFIGS, SCORER_NAMES
    from gliclass.c_amd import synth
    from gliclass.c_amd.weights import make_weights
    # normalize_features: projected text / class features L2-normalised before the scorer, logits times logit_scale (every scorer)
    cfg = dataclasses.replace(CONFIGS["tiny"], scorer=SCORER_NAMES[scorer], normalize_features=bool(normalize), logit_scale=3.5 if normalize else 1.0)
    w = make_weights(cfg, 7)
    B, S, C = 3, 48, 3
    ids, mask, _ = synth.make_inputs(cfg, B, S, C, seed=5, ragged=True)
    logits, hidden = oracle_c.forward(cfg, w, ids, mask, want_hidden=True)
    assert logits.shape == (B, C)
    H = cfg.hidden
    X = torch.tensor(hidden[-1])
    T = lambda n: torch.tensor(w[n])
    lin = lambda x, p: torch.nn.functional.linear(x, T(p + ".weight"), T(p + ".bias"))
    proj = lambda x, p: lin(torch.nn.functional.gelu(lin(x, p + ".linear_1")), p + ".linear_2")
    text = proj(X[:, 0, :], "text_projector")                                       # pooling 'first'
    pos = np.stack([np.nonzero(ids[b] == cfg.class_token_index)[0][:C] for b in range(B)])
    cls = proj(torch.stack([X[b, pos[b], :] for b in range(B)]), "classes_projector")     # [B, C, H]
    if normalize:
        text = text / (text.norm(dim=-1, keepdim=True) + 1e-8)
        cls = cls / (cls.norm(dim=-1, keepdim=True) + 1e-8)
    if scorer == "weighted-dot":
        t = lin(text, "scorer.proj_text").view(B, 1, 1, 2, H)
        l = lin(cls, "scorer.proj_label").view(B, 1, C, 2, H)
        t = t.expand(-1, -1, C, -1, -1).permute(3, 0, 1, 2, 4)
        l = l.expand(-1, 1, -1, -1, -1).permute(3, 0, 1, 2, 4)
        cat = torch.cat([t[0], l[0], t[1] * l[1]], dim=-1)
        want = lin(torch.relu(lin(cat, "scorer.out_mlp.0")), "scorer.out_mlp.3").view(B, C)
    else:
        comb = torch.cat([text.unsqueeze(1).expand(-1, C, -1), cls], dim=-1)
        want = lin(torch.relu(lin(torch.relu(lin(comb, "scorer.mlp.0")), "scorer.mlp.2")), "scorer.mlp.4").squeeze(-1)
    if normalize:
        want = want * cfg.logit_scale
    want = want.numpy()
    assert np.abs(want).max() > 0.05 and np.abs(want).max() < 30, "synthetic scorer weights should give logits in the sigmoid's range"
    np.testing.assert_allclose(logits, want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))
