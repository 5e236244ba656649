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
