#!/bin/bash
# Runs the native launcher end to end on an MI355X with files made on the spot: the fixture tokenizer (tests/golden), a random-weight
# model whose vocabulary matches it, examples/demo/data.json.  Scores are meaningless (random weights); the point is the path:
# JSON -> prompts -> C tokenizer -> HIP forward -> printed scores, exactly the reference's `./GLiClass data.json true`.
# usage: scripts/demo_launcher.sh [config=small] [dtype=f16]
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
CFG=${1:-small}; export GLICLASS_DTYPE=${2:-f16}
WORK=$(mktemp -d /tmp/glc_demo.XXXXXX)
make -s -C "$REPO/gliclass/c_amd" -j8 all
zcat "$REPO/tests/golden/tokenizer.json.gz" > "$WORK/tokenizer.json"
python3 - "$REPO" "$CFG" "$WORK/model.glcw" <<'PY'
import dataclasses, sys
sys.path.insert(0, sys.argv[1])
from gliclass.c_amd import weights
from gliclass.c_amd.config import CONFIGS
cfg = dataclasses.replace(CONFIGS[sys.argv[2]], vocab=6003, class_token_index=6001, text_token_index=6002)
weights.write_blob(sys.argv[3], cfg, weights.make_weights(cfg, 42))
PY
GLICLASS_THRESHOLD=${GLICLASS_THRESHOLD:-0.5} "$REPO/gliclass/c_amd/gliclass_main" "$REPO/examples/demo/data.json" true "$WORK/tokenizer.json" "$WORK/model.glcw"
rm -rf "$WORK"
