#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not shipped): the byte-level BPE tokenizer fixture and its golden vectors.

The decoder-style GLiClass models the reference's README names (/root/reference/Readme.md:91-94: gliclass-qwen-*, gliclass-llama-*)
ship a byte-level BPE `tokenizer.json` (Qwen2: NFC normalizer, Split(<regex>, Isolated) + ByteLevel(use_regex=false) pre-tokenizer,
BPE model, ByteLevel decoder).  No such file is on disk, so — exactly as oracle/gen_tokenizer_fixture.py does for the DeBERTa
family — a stand-in with THAT structure is trained offline (python docstrings + a multilingual sample) with the Rust `tokenizers`
library (the one tokenizers-cpp wraps, /root/reference/src/tokenizer.c:33,175), the GLiClass tokens are added
(/root/reference/src/preprocessor.c:68-69), and the library's ids for a set of probe texts become the golden vectors of the native
implementation (gliclass/c_amd/host/tokenizer.c):   tests/golden/bpe_tokenizer.json.gz, tests/golden/bpe_golden.json.gz.
Run here (needs tokenizers; no network):  python oracle/gen_bpe_fixture.py"""
import gzip
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gen_tokenizer_fixture import MULTI, english_corpus, probe_texts  # noqa: E402

QWEN_SPLIT = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"


def build(vocab_size=6000):
    from tokenizers import AddedToken, Regex, Tokenizer, decoders, normalizers, pre_tokenizers, processors, trainers
    from tokenizers.models import BPE
    tok = Tokenizer(BPE(unk_token=None, continuing_subword_prefix="", end_of_word_suffix="", fuse_unk=False, byte_fallback=False))
    tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(QWEN_SPLIT), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    tok.post_processor = processors.ByteLevel(trim_offsets=False)
    eng = english_corpus()
    rnd = random.Random(0)
    rnd.shuffle(eng)
    corpus = eng[:40000] + MULTI * 40
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=["<|endoftext|>", "<|im_start|>", "<|im_end|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(corpus, trainer)
    tok.add_tokens(["<<LABEL>>", "<<SEP>>"])              # what `tokenizer.add_tokens([...])` upstream produces
    return tok


def extra_texts():
    return ["I'm sure it's what they've said; we'll see, he'd go, you're RIGHT, DON'T, I'M, it'S, ſ's 'S 'Re 'LL", "line one\nline two\r\n\r\nline four   \n   indented",
            "trailing spaces   ", "   leading", "a  b   c    d", "tabs\t\tand  \t mixed \n\t ws", "12345 6789 3.14 1,000,000 ½ ² ٣ ४ 𝟙", "x=y+z;  foo(bar)[0] -> {a: b}  // c",
            "    def f(x):\n        return x ** 2  # comment\n\n\n", "!!!???...---___", " !leading punct", "word's 'quoted' \"double\"", "naïve café déjà vu ﬁ",
            "ẹ́ ạ́ ợ 각 Å Å", "ẛ̣ क़ ཱི ཱུ ཱྀ אָּ", "<|endoftext|>hello<|im_start|>user\nhi<|im_end|>", "mixed中文English日本語123",
            "\r\n\r\n", "\n", " \n ", "a b c　d", "​‍﻿ zero width", "end with newline\n", "Ünï 's ÜBER'S"]


def main():
    tok = build()
    js = tok.to_str()
    with gzip.GzipFile(os.path.join(GOLD, "bpe_tokenizer.json.gz"), "wb", mtime=0) as f:
        f.write(js.encode("utf-8"))
    texts = probe_texts() + extra_texts()
    gold = {"generator": "oracle/gen_bpe_fixture.py", "tokenizers_version": __import__("tokenizers").__version__, "texts": texts,
            "normalized": [tok.normalizer.normalize_str(t) for t in texts],
            "ids": [tok.encode(t, add_special_tokens=True).ids for t in texts],
            "decoded": [tok.decode(tok.encode(t, add_special_tokens=True).ids, skip_special_tokens=False) for t in texts[:80]]}
    with gzip.GzipFile(os.path.join(GOLD, "bpe_golden.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(gold, ensure_ascii=True).encode("ascii"))
    print("vocab", tok.get_vocab_size(), "texts", len(texts), "json bytes", len(js))


if __name__ == "__main__":
    main()
