#!/usr/bin/env python3
"""End-to-end launcher throughput (text in -> printed scores): gliclass_main on N synthetic texts with the fixture tokenizer and
a base-shaped model (random weights, vocabulary of the fixture tokenizer), three-phase (GLICLASS_PIPELINE=0, the reference's
structure) against the per-batch pipeline, at the reference's batch size 8 and at 64.  Prints one JSON line per setting."""
import dataclasses
import gzip
import json
import os
import random
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    from gliclass.c_amd import weights
    from gliclass.c_amd.config import CONFIGS
    n_texts = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    words = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    cname = sys.argv[3] if len(sys.argv) > 3 else "base"
    wd = tempfile.mkdtemp(prefix="glc_launch_", dir="/tmp")
    tokp = os.path.join(wd, "tokenizer.json")
    open(tokp, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer.json.gz")).read())
    cfg = dataclasses.replace(CONFIGS[cname], name=cname + "-tok", vocab=6003, class_token_index=6001, text_token_index=6002)
    blob = os.path.join(wd, "model.glcw")
    weights.write_blob(blob, cfg, weights.make_weights(cfg, 42))
    gold = json.loads(gzip.open(os.path.join(ROOT, "tests", "golden", "tokenizer_golden.json.gz")).read())
    vocab = " ".join(gold["texts"][40:54]).split()
    rnd = random.Random(1)
    texts = [" ".join(rnd.choice(vocab) for _ in range(rnd.randint(words // 2, words))) for _ in range(n_texts)]
    data = os.path.join(wd, "data.json")
    json.dump({"texts": texts, "labels": [["travel", "dreams", "sport", "science", "politics", "economy", "health", "art"]],
               "same_labels": True, "classification_type": "multi-label"}, open(data, "w"))
    exe = os.path.join(ROOT, "gliclass", "c_amd", "gliclass_main")
    for bs in ("8", "64"):
        for pipeline in ("0", "1"):
            env = dict(os.environ, GLICLASS_PIPELINE=pipeline, GLICLASS_BATCH_SIZE=bs, GLICLASS_MAX_LENGTH="1024")
            best = None
            for rep in range(2):
                t0 = time.time()
                r = subprocess.run([exe, data, "true", tokp, blob], capture_output=True, text=True, env=env)
                wall = time.time() - t0
                if r.returncode != 0:
                    print(r.stderr[-2000:])
                    sys.exit(1)
                m = re.search(r"Execution time: ([0-9.]+) seconds", r.stdout)
                stage = float(m.group(1))
                best = stage if best is None else min(best, stage)
            print(json.dumps({"model": cfg.name, "texts": n_texts, "avg_words": words * 0.75, "batch_size": int(bs), "pipeline": int(pipeline),
                              "stages_s": round(best, 4), "texts_per_s": round(n_texts / best, 1), "wall_s_incl_load": round(wall, 2),
                              "blocks": r.stdout.count("Text_")}), flush=True)


if __name__ == "__main__":
    main()
