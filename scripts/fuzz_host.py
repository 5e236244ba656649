#!/usr/bin/env python3
"""Mutation fuzz of the host layer's parsers (CPU; run under the ASan build: see scripts/asan_host.sh for the LD_PRELOAD line):
tokenizers_new_from_str on mutated tokenizer.json documents, parse_json on mutated input documents, glc_weights_load on
mutated safetensors headers / config.json.  A crash or a sanitizer report is the failure; wrong documents must be refused."""
import ctypes as C
import gzip
import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gliclass.c_amd import _lib  # noqa: E402


def mutate(b, rnd):
    b = bytearray(b)
    for _ in range(rnd.randint(1, 6)):
        k = rnd.random()
        if not b:
            break
        i = rnd.randrange(len(b))
        if k < 0.3:
            b[i] = rnd.randrange(256)
        elif k < 0.5:
            del b[i:i + rnd.randint(1, 40)]
        elif k < 0.7:
            b[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 8)))
        elif k < 0.85:
            b[i:i] = rnd.choice([b'{', b'}', b'[', b']', b'"', b',', b':', b'\\u', b'\\', b'null', b'1e999', b'-', b'\x00'])
        else:
            j = rnd.randrange(len(b))
            b[i:i + 20] = b[j:j + 20]
    return bytes(b)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    L = _lib.model()
    js = json.loads(gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer.json.gz")).read())
    js["model"]["vocab"] = js["model"]["vocab"][:300] + js["model"]["vocab"][-1:]          # small document: mutations hit structure, not vocabulary
    for a in js["added_tokens"]:
        if a["id"] >= 300:
            a["id"] = 300
    small = json.dumps(js).encode()
    ok = bad = 0
    probe = "<<LABEL>>a b<<SEP>> héllo ①  x".encode()
    res = _lib.TokenizerEncodeResult()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)                                   # the refusals are noisy by design
    try:
        for _ in range(n):
            doc = mutate(small, rnd)
            h = L.tokenizers_new_from_str(doc, len(doc))
            if h:
                ok += 1
                L.tokenizers_encode(h, probe, len(probe), 1, C.byref(res))
                libc.free(res.token_ids)
                L.tokenizers_free(h)
            else:
                bad += 1
        data = json.dumps({"texts": ["a", "b"], "labels": [["x", "y"], ["z"]], "same_labels": False, "classification_type": "multi-label"}).encode()
        for _ in range(n):
            doc = mutate(data, rnd).replace(b"\x00", b" ")
            texts, labels, nl = C.POINTER(C.c_char_p)(), C.POINTER(C.POINTER(C.c_char_p))(), C.POINTER(C.c_size_t)()
            nt, nls, same, ct = C.c_size_t(0), C.c_size_t(0), C.c_bool(False), C.c_char_p()
            L.parse_json(doc, C.byref(texts), C.byref(nt), C.byref(labels), C.byref(nl), C.byref(nls), C.byref(same), C.byref(ct))
        with tempfile.TemporaryDirectory() as d:
            cfg = {"model_type": "GLiClass", "encoder_config": {"model_type": "deberta-v2", "hidden_size": 128, "num_hidden_layers": 1,
                                                                "num_attention_heads": 2, "intermediate_size": 256, "relative_attention": True,
                                                                "position_buckets": 256, "max_position_embeddings": 512, "norm_rel_ebd": "layer_norm",
                                                                "share_att_key": True, "pos_att_type": "p2c|c2p", "position_biased_input": False,
                                                                "type_vocab_size": 0}}
            hdr = json.dumps({"embeddings.word_embeddings.weight": {"dtype": "F32", "shape": [8, 128], "data_offsets": [0, 4096]}}).encode()
            W = _lib.Weights()
            for _ in range(n // 3):
                open(os.path.join(d, "config.json"), "wb").write(mutate(json.dumps(cfg).encode(), rnd))
                h2 = mutate(hdr, rnd)
                open(os.path.join(d, "model.safetensors"), "wb").write(len(h2).to_bytes(8, "little") + h2 + b"\0" * 4096)
                if L.glc_weights_load(d.encode(), C.byref(W)) == 0:
                    L.glc_weights_free(C.byref(W))
    finally:
        os.dup2(saved, 2)
    print(f"tokenizer documents: {ok} accepted, {bad} refused; parse_json and checkpoint mutations survived; no crash")


if __name__ == "__main__":
    main()
