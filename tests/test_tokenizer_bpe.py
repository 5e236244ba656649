"""CPU: the byte-level BPE front end of the native tokenizer (NFC normalizer, the GPT-2 / Qwen2 split patterns, byte-level mapping,
BPE merges, ByteLevel decoder — gliclass/c_amd/host/tokenizer.c) against HF `tokenizers`, the Rust library the reference calls
through tokenizers-cpp (/root/reference/src/tokenizer.c:33,175).  This is the tokenizer family of the decoder-style models the
reference's README names (Readme.md:91-94).  Golden ids: oracle/gen_bpe_fixture.py; with the `tokenizers` wheel importable the
comparison is repeated live on random text, on the GPT-2 form of the pre-tokenizer and on added-token variants."""
import ctypes as C
import gzip
import json
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def tk_json():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "gliclass", "c_amd"), "-j4", "libgliclass_model.so"], stdout=subprocess.DEVNULL)
    return gzip.open(os.path.join(GOLD, "bpe_tokenizer.json.gz")).read().decode("utf-8")


@pytest.fixture(scope="module")
def gold():
    return json.loads(gzip.open(os.path.join(GOLD, "bpe_golden.json.gz")).read())


@pytest.fixture(scope="module")
def tok(tk_json):
    from gliclass.c_amd.tokenizer import Tokenizer
    return Tokenizer(tk_json)


def test_nfc_matches_golden(tok, gold):
    for t, want in zip(gold["texts"], gold["normalized"]):
        assert tok.normalize(t) == want, repr(t)


def test_bpe_ids_match_golden(tok, gold):
    assert len(gold["texts"]) >= 250
    for t, want in zip(gold["texts"], gold["ids"]):
        assert tok.encode(t, True) == want, repr(t)


def test_bytelevel_decode_matches_golden(tok, gold):
    for t, ids, want in zip(gold["texts"], gold["ids"], gold["decoded"]):
        assert tok.decode(ids, skip_special_tokens=False) == want, repr(t)


def test_decoder_prompt_through_tokenize_inputs(tok, gold):
    """The reference's batch path (tokenize_inputs: pad to the longest, raw cut at max_length) on GLiClass prompts."""
    texts = [t for t in gold["texts"] if "<<LABEL>>" in t][:16]
    ids, mask = tok.tokenize_inputs(texts, 64)
    S = min(64, max(len(tok.encode(t, True)) for t in texts))
    assert all(len(r) == S for r in ids)
    for t, r, m in zip(texts, ids, mask):
        want = tok.encode(t, True)[:S]
        assert r[: len(want)] == want and m[: len(want)] == [1] * len(want) and all(v == 0 for v in m[len(want):])
    lab = tok.token_to_id("<<LABEL>>")
    assert lab >= 6000 and any(lab in r for r in ids)


def _rand_text(rnd):
    pools = [(0x20, 0x7f)] * 6 + [(0xa0, 0x250), (0x300, 0x370), (0x370, 0x400), (0x400, 0x500), (0x590, 0x700), (0x900, 0x980), (0xe00, 0xe80),
                                  (0x1100, 0x1200), (0x1e00, 0x2000), (0x2000, 0x2070), (0x3000, 0x3100), (0x4e00, 0x4f00), (0xac00, 0xad00),
                                  (0xfb00, 0xfb50), (0xff00, 0xfff0), (0x1f300, 0x1f650), (0x1d400, 0x1d500)]
    s = []
    for _ in range(rnd.randint(1, 60)):
        r = rnd.random()
        if r < 0.25:
            s.append(rnd.choice([" ", "  ", "\n", "\r\n", "\t", "'s", "'LL", "'t ", " 12", "3", "!?", "...", " -", "_", "́", "̈", "゙", " the", " and"]))
        else:
            lo, hi = rnd.choice(pools)
            c = rnd.randrange(lo, hi)
            s.append(chr(c if not 0xd800 <= c < 0xe000 else 0x41))
    return "".join(s)


def test_live_against_rust_tokenizers(tk_json):
    tokenizers = pytest.importorskip("tokenizers")
    from gliclass.c_amd.tokenizer import Tokenizer
    base = json.loads(tk_json)
    variants = {"qwen": base}
    gpt2 = json.loads(tk_json)                       # GPT-2 form: ByteLevel alone with its own regex, with and without the prefix space
    gpt2["normalizer"] = None
    gpt2["pre_tokenizer"] = {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": True}
    variants["gpt2"] = gpt2
    g2 = json.loads(json.dumps(gpt2)); g2["pre_tokenizer"]["add_prefix_space"] = True
    variants["gpt2-prefix-space"] = g2
    sp = json.loads(tk_json)                         # <<LABEL>> / <<SEP>> registered as special, not normalised
    for a in sp["added_tokens"]:
        if a["content"] in ("<<LABEL>>", "<<SEP>>"):
            a.update(normalized=False, special=True)
    variants["special-labels"] = sp
    im = json.loads(tk_json); im["model"]["ignore_merges"] = True
    variants["ignore-merges"] = im
    rnd = random.Random(20261003)
    texts = [_rand_text(rnd) for _ in range(400)] + ["<<LABEL>>a b<<LABEL>>c<<SEP>> " + _rand_text(rnd) for _ in range(40)]
    # long pre-tokens (no whitespace): the heap merge of bpe_word (pre-tokens beyond 48 symbols) must pop pairs in the Rust library's order
    letters = "etaoinshrdlucmfwypvbgkqjxzETAOIN"
    texts += ["".join(rnd.choice(letters) for _ in range(n)) for n in (49, 50, 97, 300, 2000)]
    texts += [" " + "".join(rnd.choice("theandingersto") for _ in range(700)) + " end", "a" * 5000, "ab" * 1500 + "c", "é" * 400 + "x" * 64]
    for name, js in variants.items():
        s = json.dumps(js)
        ref = tokenizers.Tokenizer.from_str(s)
        nat = Tokenizer(s)
        for t in texts:
            assert nat.encode(t, True) == ref.encode(t, add_special_tokens=True).ids, (name, t)
        nat.close()


def test_unsupported_bpe_configurations_are_refused(tk_json):
    from gliclass.c_amd.tokenizer import Tokenizer
    js = json.loads(tk_json)
    js["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"] = r"\w+"
    with pytest.raises(ValueError):
        Tokenizer(json.dumps(js))
    js = json.loads(tk_json); js["model"]["byte_fallback"] = True
    with pytest.raises(ValueError):
        Tokenizer(json.dumps(js))
