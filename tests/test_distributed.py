"""CPU, world_size 2 over gloo: the multi-process contract of bench.py (barrier + max-over-ranks timing,
whole-job aggregation) and the contiguous batch split of SURVEY.md §8e."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import time
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (1 + 2 * rank))       # rank 1 is 3x slower: the max must win

    def max_fn(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    el = bench.time_steps(step, lambda: None, dist.barrier, max_fn, steps=5, warmup=2)
    lo, hi = bench.shard_rows(13, world, rank)
    rows = torch.zeros(13)
    rows[lo:hi] = rank + 1
    dist.all_reduce(rows)
    q.put((rank, el, len(calls), (lo, hi), rows.tolist()))
    dist.destroy_process_group()


def test_two_rank_timing_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, e0, n0, s0, rows0), (r1, e1, n1, s1, rows1) = res
    assert n0 == n1 == 7                       # 2 warm-up + exactly 5 timed steps on every rank
    assert abs(e0 - e1) < 1e-9 and e0 >= 5 * 0.03 * 0.9      # identical max-over-ranks, set by the slow rank
    assert s0 == (0, 6) and s1 == (6, 13)
    assert rows0 == [1.0] * 6 + [2.0] * 7      # every row owned by exactly one rank


def test_shard_rows_partition():
    import bench
    for n in (1, 7, 64, 256):
        for world in (1, 2, 4, 8):
            spans = [bench.shard_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_kernel_flops_sum_to_survey_formula():
    import bench
    from gliclass.c_amd.config import CONFIGS
    cfg = CONFIGS["base"]
    fl = bench.kernel_flops(cfg, 64, 1024)
    per_layer = sum(fl.values())
    total = cfg.layers * per_layer + 64 * 8 * cfg.hidden ** 2 * (1 + 8)
    assert abs(total - 64 * cfg.flops_per_seq(1024, 8)) / total < 1e-12
    assert abs(cfg.flops_per_seq(1024, 8) / 1e9 - 231.95) < 0.05          # 231.95 GFLOP / sequence


def _run_bench_stub(*flags):
    """`python bench.py --gpus N --stub ...` with NO torch.distributed environment: bench.py itself starts the N ranks (the path the
    driver's `python bench.py --gpus N` takes), joins them over gloo and runs the real partition / timing / gather code with a CPU
    stand-in for the engine step."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", *flags], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line, the other ranks nothing
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks_strong_scaling():
    """Strong mode: one global batch of 13 rows over 3 ranks (uneven shards 4/4/5), logits gathered to rank 0 every step."""
    out = _run_bench_stub("--gpus", "3", "--scaling", "strong", "--batch", "13", "--labels", "4", "--steps", "5", "--warmup", "2")
    assert out["n_gpus"] == 3 and out["scaling"] == "strong" and out["global_batch"] == 13
    assert out["rows_rank0"] == 4 and out["step_calls_rank0"] == 7
    assert out["gathered_rows"] == list(range(13))          # every row exactly once, in order, on rank 0
    assert out["value"] > 0


def test_bench_spawns_its_own_ranks_weak_scaling():
    out = _run_bench_stub("--gpus", "2", "--scaling", "weak", "--batch", "6", "--labels", "2", "--steps", "4", "--warmup", "1")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["global_batch"] == 12 and out["rows_rank0"] == 6
    assert out["step_calls_rank0"] == 5


def test_bench_refuses_more_ranks_than_rows_and_relays_failure():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "3", "--scaling", "strong", "--batch", "2"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "got no rows" in r.stderr


def test_bench_without_gpus_fails_loudly():
    """No GPU in this container: the real (non-stub) multi-rank launch must refuse before starting any rank."""
    import subprocess
    import torch as _t
    if _t.cuda.device_count() >= 2:
        pytest.skip("GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr


def test_bench_c4_alias_is_the_strong_batch_shard():
    """`bench.py --config c4 --gpus N` = BASELINE.json configs[3] as one flag: gliclass-large, ONE global batch of 256 split over the
    ranks (32 per GPU at N = 8), logits gathered to rank 0 (the reference's batch loop, /root/reference/main.c:141-150, as a shard)."""
    out = _run_bench_stub("--gpus", "2", "--config", "c4", "--labels", "2", "--steps", "2", "--warmup", "1")
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["global_batch"] == 256 and out["rows_rank0"] == 128
    assert out["gathered_rows"] == list(range(256))


def test_bench_eight_ranks_c4_shard_rehearsal():
    """The driver's 8-GPU run, rehearsed on the CPU: `bench.py --gpus 8 --config c4` starts 8 ranks itself, joins them
    (gloo here, RCCL there), gives each its 32 contiguous rows of the global batch of 256 and gathers the logits in order on rank 0 every
    step; the line carries every rank's own step time (min / max: a straggler is visible) and the host threads each rank was given —
    at least one even when the CPU quota is smaller than the rank count."""
    out = _run_bench_stub("--gpus", "8", "--config", "c4", "--labels", "2", "--steps", "3", "--warmup", "1")
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["global_batch"] == 256 and out["rows_rank0"] == 32
    assert out["gathered_rows"] == list(range(256))
    assert out["step_calls_rank0"] == 4
    assert out["omp_threads_per_rank"] >= 1
    assert 0 < out["rank_ms_per_step"]["min"] <= out["rank_ms_per_step"]["max"]
