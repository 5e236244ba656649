"""GPU (-m gpu): the RCCL branch of bench.py on real hardware at world size 1 — process-group init over the `nccl` backend (= RCCL on
ROCm), device barrier, max-over-ranks all-reduce and the per-step all-gather of the logits — in a FRESH child process (a process that
has touched the GPU is never re-executed).  The multi-rank partition / gather logic itself is covered on CPU over gloo
(tests/test_distributed.py); the N > 1 hardware run is the driver's.  Replaces /root/reference/main.c:141-150 (the batch loop)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, name, force_dist):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GLC_BENCH_FORCE_DIST")}
    if force_dist:
        env["GLC_BENCH_FORCE_DIST"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dump = str(tmp_path / f"{name}.npy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong", "--config", "small", "--batch", "8",
                        "--seq", "128", "--steps", "2", "--warmup", "1", "--cpu-seqs", "0", "--no-profile", "--dump-logits", dump],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), np.load(dump)


def test_rccl_path_at_world_size_one(tmp_path):
    plain, l0 = _bench(tmp_path, "plain", False)
    rccl, l1 = _bench(tmp_path, "rccl", True)
    assert plain["n_gpus"] == rccl["n_gpus"] == 1 and rccl["scaling"] == "strong"
    assert "gather_ms" in rccl and rccl["gather_ms"] > 0          # the all-gather really ran (RCCL, world size 1)
    assert plain["parity_ok"] is None and rccl["parity_ok"] is None   # oracle leg skipped: no parity claim on these lines
    assert l0.shape == l1.shape == (8, 8) and np.isfinite(l1).all()
    assert np.array_equal(l0, l1)                                  # gathered logits == the engine's own, bit for bit
