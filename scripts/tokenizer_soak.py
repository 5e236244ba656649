#!/usr/bin/env python3
"""Wide-Unicode soak of the native tokenizer against the Rust `tokenizers` library (needs the wheel; CPU only).
usage: tokenizer_soak.py [seed] [n_texts]   -> prints the mismatches (none expected) and a count."""
import gzip
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SPACES = [" ", "\t", "\n", "\r", " ", " ", "​", "‍", "﻿", "　", " ", ""]
MARKS = ["‍", "️", "⃣", "\U0001f3fb", "\U0001f3ff", "्", "ः", "؀", "ำ", "ᅠ", "ᆨ"]
ADDED = ["<<LABEL>>", "<<SEP>>", "[SEP]", "[MASK]", "[CLS]", "[UNK]", "[PAD]"]


def main():
    import tokenizers as hf
    from gliclass.c_amd.tokenizer import Tokenizer
    js = gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer.json.gz")).read().decode()
    mine, ref = Tokenizer(js), hf.Tokenizer.from_str(js)
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_texts = int(sys.argv[2]) if len(sys.argv) > 2 else 20000

    def rch():
        r = rnd.random()
        if r < 0.35:
            return chr(rnd.randrange(0x20, 0x7f))
        if r < 0.45:
            return rnd.choice(SPACES)
        if r < 0.55:
            return chr(rnd.randrange(0x300, 0x370))
        if r < 0.60:
            return rnd.choice(MARKS)
        while True:
            c = rnd.randrange(0x80, 0x30000) if r < 0.9 else rnd.randrange(0x1f000, 0x1fb00)
            if not 0xd800 <= c < 0xe000:
                return chr(c)

    bad, t0 = 0, time.time()
    for _ in range(n_texts):
        n = rnd.randint(0, 24)
        s = "".join(rch() for _ in range(n))
        if rnd.random() < 0.2:
            s = s[: n // 2] + rnd.choice(ADDED) + s[n // 2:]
        a, b = mine.encode(s), ref.encode(s).ids
        if a != b:
            bad += 1
            if bad <= 10:
                print("MISMATCH", [hex(ord(c)) for c in s], "\n mine", a, "\n ref ", b, "\n norm mine", ascii(mine.normalize(s)),
                      "\n norm ref ", ascii(ref.normalizer.normalize_str(s)))
    print("texts", n_texts, "mismatches", bad, "%.1fs" % (time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
