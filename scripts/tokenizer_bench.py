#!/usr/bin/env python3
"""CPU: throughput of the native tokenizer against the Rust `tokenizers` library on the same texts (ids must be equal).
Prints texts/s and tokens/s for 1, 2, 4, ... OpenMP threads (C-level timing of tokenizers_encode_batch) and the Rust numbers."""
import ctypes as C
import gzip
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd import _lib  # noqa: E402
from gliclass.c_amd.hostinfo import effective_cpus  # noqa: E402
from gliclass.c_amd.tokenizer import Tokenizer  # noqa: E402


def main():
    js = gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer.json.gz")).read().decode()
    gold = json.loads(gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer_golden.json.gz")).read())
    rnd = random.Random(0)
    base = gold["texts"][40:54] + gold["texts"][7:9]
    texts = ["<<LABEL>>travel<<LABEL>>dreams<<LABEL>>sport<<SEP>>" + " ".join(rnd.choice(base) for _ in range(40)) for _ in range(1024)]
    mine = Tokenizer(js)
    L = _lib.model()
    bs = [t.encode() for t in texts]
    n = len(bs)
    arr = (C.c_char_p * n)(*bs)
    lens = (C.c_size_t * n)(*[len(b) for b in bs])
    res = (_lib.TokenizerEncodeResult * n)()
    omp = C.CDLL("libgomp.so.1")
    threads = [t for t in (1, 2, 4, 8, 16, 32) if t <= effective_cpus()]
    for thr in threads:
        omp.omp_set_num_threads(thr)
        t0 = time.perf_counter()
        L.tokenizers_encode_batch(mine.handle, arr, lens, n, 1, res)
        dt = time.perf_counter() - t0
        tot = sum(res[i].len for i in range(n))
        L.tokenizers_free_encode_results(res, n)
        print(f"native C, {thr:2d} thread(s): {n / dt:8.0f} texts/s  {tot / dt / 1e6:6.2f} M tokens/s  ({tot / n:.0f} tokens/text)")
    try:
        import tokenizers as hf
    except ImportError:
        return
    ref = hf.Tokenizer.from_str(js)
    t0 = time.perf_counter()
    ids = [ref.encode(t).ids for t in texts[:128]]
    dt = time.perf_counter() - t0
    print(f"Rust tokenizers, 1 thread : {128 / dt:8.0f} texts/s  {sum(map(len, ids)) / dt / 1e6:6.2f} M tokens/s")
    t0 = time.perf_counter()
    enc = ref.encode_batch(texts)
    dt = time.perf_counter() - t0
    print(f"Rust tokenizers, encode_batch (rayon): {n / dt:8.0f} texts/s")
    assert [e.ids for e in enc[:64]] == mine.encode_batch(texts[:64]), "ids differ"


if __name__ == "__main__":
    main()
