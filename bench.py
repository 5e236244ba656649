#!/usr/bin/env python3
"""Headline benchmark: sequences/s of the GLiClass forward (the work behind run_inference(),
/root/reference/src/model.c:122-207) at BASELINE.json's config c3 — gliclass-base shape, batch 64,
seq 1024, 8 labels — one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|f16|bf16] [--config base|small|large|qwen-1.5b|c2|c3|c4|c5]
                  [--batch B] [--seq S] [--labels C] [--scaling weak|strong]
  (--config c2 .. c5 = BASELINE.json's configs as one flag: c4 = large, one global batch of 256 split over the ranks, strong scaling)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launched WITHOUT a torch.distributed environment and with --gpus N > 1, this script starts the N ranks itself (fresh child
processes, created before the parent touches the GPU), so `python bench.py --gpus 8` is a complete 8-GPU run.

A step = one forward over one batch whose token ids / mask are already resident in HBM (glc_engine_forward_device).
  --scaling weak   (default) every rank owns a full batch of --batch rows; no data-path collective (sequences are independent,
                   SURVEY.md §8e); value = N*B*K / max-over-ranks time.
  --scaling strong ONE global batch of --batch rows is partitioned contiguously over the ranks (the reference's batch loop,
                   /root/reference/main.c:141-150, turned into a batch shard: c4 = `--config large --batch 256 --gpus 8` is 32 rows per
                   GPU) and every step ends with the gather of the [B/G, C] logits to rank 0 (RCCL all-gather); value = B*K / time.

The headline mode is the DEFAULT arithmetic mode of the product (GLICLASS_DTYPE unset = f32: fp32 data, every matrix product as
split-f16 MFMAs), the mode that meets the 1e-3 probability tolerance; rank 0 checks it live against the CPU oracle and prints
`parity_ok`.  The opt-in 16-bit throughput mode is measured beside it (`throughput_mode`), flagged with its own error.
Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed on the engine's stream) and `cpu_baseline`
(the C oracle — a port, not ONNXRuntime — on a bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 2500.0}   # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md); the f32 mode runs f16 MFMAs too (x3 per product)
MFMA_PER_PRODUCT = {"f16": 1, "bf16": 1, "f32": 3}            # split-f16: a_lo*b_hi + a_hi*b_lo + a_hi*b_hi (MX projections: 1 f16 + 1 fp8 at twice the rate = 2, profile_mode)
BAR = 1e-3                                                     # /root/reference/ONNX_CONVERTING/test_onnx.py:30


def kernel_flops(cfg, B, S):
    """Algorithmic FLOPs of ONE launch of each kernel class (SURVEY.md §8d terms)."""
    H, I, P = cfg.hidden, cfg.inter, 2 * cfg.att_span
    M = B * S
    if cfg.backbone == 1:          # decoder-style backbone (Qwen2 arithmetic)
        nqd, nkvd = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
        kappa = 0.5 if cfg.causal else 1.0
        return {
            "gemm_qkv": 2.0 * M * H * (nqd + 2 * nkvd),
            "attention": B * S * 4.0 * S * nqd * kappa,
            "gemm_attn_out": 2.0 * M * nqd * H,
            "gemm_ffn1_gelu": 2.0 * M * H * 2 * I,
            "gemm_ffn2": 2.0 * M * I * H,
        }
    return {
        "gemm_qkv": 2.0 * M * H * 3 * H,
        "attention": B * S * (4.0 * S * H + 4.0 * P * H),
        "gemm_attn_out": 2.0 * M * H * H,
        "gemm_ffn1_gelu": 2.0 * M * H * I,
        "gemm_ffn2": 2.0 * M * I * H,
    }


def git_head():
    """The commit this tree is at (None where no .git travels with it, e.g. on a gpurun box: scripts pass GLICLASS_BENCH_COMMIT there)."""
    try:
        head = open(os.path.join(ROOT, ".git", "HEAD")).read().strip()
        if head.startswith("ref:"):
            return open(os.path.join(ROOT, ".git", head[5:])).read().strip()
        return head
    except Exception:
        return None


def time_steps(step_fn, sync_fn, barrier_fn, max_fn, steps, warmup, own=None):
    """The timing contract: W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on
    both sides; returns the max over ranks of the elapsed seconds.  `own` (a list) receives this rank's own
    time to its device sync, before the closing barrier: a straggler shows in the spread of those."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    barrier_fn()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    t_own = time.perf_counter()
    barrier_fn()
    t1 = time.perf_counter()
    if own is not None:
        own.append(t_own - t0)
    return max_fn(t1 - t0)


def shard_rows(n_rows, world, rank):
    """Contiguous batch split of the strong mode: rank g gets rows [g*n/G, (g+1)*n/G)."""
    lo = n_rows * rank // world
    hi = n_rows * (rank + 1) // world
    return lo, hi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, env_extra=None, timeout_s=3000):
    """Start the N ranks of one node as fresh child processes (the parent has not touched the GPU: nothing here initialises HIP)
    and relay their exit status.  A rank that dies takes the job down: the others are terminated by PID."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    deadline = time.time() + timeout_s
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
        if (rc != 0 or time.time() > deadline) and live:
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            live = []
    return rc


# ---------------------------------------------------------------------------------------------------------------------------
# runners: what a rank executes per step.  EngineRunner = the product (HIP engine through the C-ABI); StubRunner = a CPU stand-in
# with the same interface so that tests/test_distributed.py can drive the REAL multi-rank code below over gloo without a GPU.
class EngineRunner:
    def __init__(self, args, cfg, W, local_rank, dtype, ids, mask):
        import torch
        from gliclass.c_amd import _lib
        from gliclass.c_amd.engine import DTYPES
        self.torch = torch
        self.hipl = _lib.hip()
        self.B, self.S = ids.shape
        self.Cn = args.labels
        self.h = self.hipl.glc_engine_create(C.byref(W.cfg), C.cast(W.tensors, C.POINTER(C.c_void_p)), W.n_tensors, local_rank, DTYPES[dtype])
        if not self.h:
            raise SystemExit("bench: engine create failed: " + self.hipl.glc_last_error().decode())
        self.d_ids = self.hipl.glc_device_malloc(self.h, ids.nbytes)
        self.d_mask = self.hipl.glc_device_malloc(self.h, mask.nbytes)
        # logits live in a torch tensor so that the strong mode can hand them to RCCL without a copy
        self.logits = torch.zeros((self.B, self.Cn), dtype=torch.float32, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        self.hipl.glc_memcpy_h2d(self.h, self.d_ids, ids.ctypes.data, ids.nbytes)
        self.hipl.glc_memcpy_h2d(self.h, self.d_mask, mask.ctypes.data, mask.nbytes)

    def step(self):
        if self.hipl.glc_engine_forward_device(self.h, self.d_ids, self.d_mask, self.B, self.S, self.Cn, C.c_void_p(self.logits.data_ptr())) != 0:
            raise RuntimeError(self.hipl.glc_last_error().decode())

    def sync(self):
        if self.hipl.glc_engine_sync(self.h) != 0:
            raise RuntimeError(self.hipl.glc_last_error().decode())
        self.torch.cuda.synchronize()

    def close(self):
        self.sync()
        self.hipl.glc_device_free(self.h, self.d_ids)
        self.hipl.glc_device_free(self.h, self.d_mask)
        self.hipl.glc_engine_destroy(self.h)
        self.h = None


class StubRunner:
    """logits[b, j] = global row id * 16 + j — lets a CPU test check that the gather puts every row where it belongs."""
    def __init__(self, args, row_lo, n_rows):
        import torch
        self.torch = torch
        self.row_lo, self.B, self.Cn = row_lo, n_rows, args.labels
        self.logits = torch.zeros((n_rows, self.Cn), dtype=torch.float32)
        self.calls = 0

    def step(self):
        self.calls += 1
        r = self.torch.arange(self.row_lo, self.row_lo + self.B, dtype=self.torch.float32)[:, None]
        self.logits.copy_(r * 16 + self.torch.arange(self.Cn, dtype=self.torch.float32)[None, :])
        time.sleep(0.002)

    def sync(self):
        pass

    def close(self):
        pass


def gather_logits(dist, runner, world, rank, global_rows, Cn):
    """Strong mode, once per step: the [rows_g, C] logits of every rank -> rank 0's [B, C] (a < 10 KB, latency-bound
    all-gather; the only collective on the data path).  Shards are padded to the largest one (B need not divide by G)."""
    torch = runner.torch
    if dist is None:
        return runner.logits
    cap = max(shard_rows(global_rows, world, r)[1] - shard_rows(global_rows, world, r)[0] for r in range(world))
    mine = runner.logits
    if mine.shape[0] < cap:
        mine = torch.cat([mine, torch.zeros((cap - mine.shape[0], Cn), dtype=mine.dtype, device=mine.device)])
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine.contiguous())
    if rank != 0:
        return None
    return torch.cat([parts[r][: shard_rows(global_rows, world, r)[1] - shard_rows(global_rows, world, r)[0]] for r in range(world)])


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, wptrs, S, C_labels, seqs, seqs8):
    """The CPU oracle timed on this box's host cores: all the cores this job may use, then OMP_NUM_THREADS=8 — the reference's
    NUM_THREADS (/root/reference/include/configs.h:7) — on a smaller sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    from gliclass.c_amd import synth
    ids, mask, _ = synth.make_inputs(cfg, seqs, S, C_labels, seed=1234)
    lib = oracle_c.lib()              # team sized to the CPUs this job may really use (affinity / cgroup quota): `cores` below
    cc = oracle_c._cfg(cfg)
    ptrs = C.cast(wptrs, C.POINTER(C.c_void_p))
    fwd = lib.glo_forward_decoder if cfg.backbone == 1 else lib.glo_forward

    def run(n):
        logits = np.zeros((n, C_labels), np.float32)
        c_out = C.c_int(0)
        a = [C.byref(cc), ptrs, ids.ctypes.data, mask.ctypes.data, n, S, logits.ctypes.data, C_labels, C.byref(c_out), None]
        if cfg.backbone != 1:
            a.append(None)
        t0 = time.perf_counter()
        rc = fwd(*a)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("oracle failed")
        return dt, logits
    cores = int(lib.glo_num_threads())
    dt, logits = run(seqs)
    out = dict(value=seqs / dt, unit="sequences/s", cores=cores, kind="port", cpu_model=cpu_model_name(),
               sample=f"{seqs} sequence(s) of the same workload (S={S}, {C_labels} labels), fp32 C/OpenMP oracle "
                      f"(CPU restatement, not ONNXRuntime), {dt:.1f} s")
    if seqs8 > 0 and cores > 8:
        lib.glo_set_threads(8)
        dt8, _ = run(min(seqs8, seqs))
        lib.glo_set_threads(cores)
        out["value_8_threads"] = min(seqs8, seqs) / dt8
        out["sample_8_threads"] = f"{min(seqs8, seqs)} sequence(s), OMP_NUM_THREADS=8 (the reference's NUM_THREADS), {dt8:.1f} s"
    return out, logits, ids, mask


def profile_mode(hipl, h, step, sync, cfg, B, S, Cn, dtype, seqs_per_s_one_gpu, config_key, mx=False, mx_attn=False):
    """Per-kernel-class HIP-event profile of 3 forwards on the engine's stream -> the `roofline` object."""
    fl = kernel_flops(cfg, B, S)
    hipl.glc_profile_enable(h, 1)
    for _ in range(3):
        step()
        sync()
    names = (C.c_char_p * 16)(); ms = (C.c_float * 16)(); cnt = (C.c_int * 16)()
    k = hipl.glc_profile_read(h, names, ms, cnt, 16)
    hipl.glc_profile_enable(h, 0)
    prof = {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(k) if cnt[i] > 0}
    total_ms = sum(v[0] for v in prof.values())
    per = {}
    for n, (tms, c) in prof.items():
        avg = tms / c
        e = dict(avg_ms=round(avg, 4), launches_per_fwd=c // 3, share=round(tms / total_ms, 4))
        if n in fl:
            e["tflops"] = round(fl[n] / (avg * 1e-3) / 1e12, 1)
        per[n] = e
    dom = max((n for n in per if n in fl), key=lambda n: per[n]["avg_ms"] * per[n]["launches_per_fwd"])
    peak = PEAK_TFLOPS[dtype]
    e2e = cfg.flops_per_seq(S, Cn) * seqs_per_s_one_gpu / 1e12
    # FLOPs really executed per forward: the last layer runs Q / attention output / FFN only on the B*(1+C) rows the head reads
    executed = e2e
    if cfg.backbone != 1 and os.environ.get("GLICLASS_PRUNE_LAST", "1") != "0":
        H, I, P = cfg.hidden, cfg.inter, 2 * cfg.att_span
        full_last = S * (8.0 * H * H + 4.0 * H * I + 4.0 * S * H + 4.0 * P * H)
        kept_last = S * 4.0 * H * H + (1 + Cn) * (4.0 * H * H + 4.0 * H * I + 4.0 * S * H + 4.0 * P * H)      # K, V for every row; the rest on 1+C rows
        executed = (cfg.flops_per_seq(S, Cn) - full_last + kept_last) * seqs_per_s_one_gpu / 1e12
    traffic, traffic_src = None, None
    try:        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (cannot be collected live)
        ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(dtype)
        if ent and ent.get("config") == config_key and dom in ent.get("kernels", {}):
            traffic = ent["kernels"][dom]["hbm_bytes_per_launch"]
            traffic_src = "profiles/traffic.json (" + ent.get("source", "?") + (", commit " + str(ent["commit"])[:12] if ent.get("commit") else "") + ")"
    except Exception:
        pass
    # matrix-pipe time per product in f16-MFMA units, per kernel class.  MX (projections; attention when it runs on MX tiles): a_hi*w_hi as
    # f16 MFMAs + both cross terms as one block-scaled fp8 MFMA at twice the f16 rate = 2; split-f16 units = 3; 16-bit modes = 1
    def units(name):
        if dtype != "f32":
            return MFMA_PER_PRODUCT[dtype]
        if name.startswith("gemm"):
            return 2 if mx else 3
        if name == "attention":
            return 2 if (mx and mx_attn) else 3
        return 3
    mpp = units(dom)
    # whole-forward matrix-pipe occupancy: every class at its own unit count, weighted by its algorithmic FLOPs per forward
    # (the pruned last layer and the head run split-f16 units; they are < 3 % of the FLOPs)
    w_fl = sum(fl[n] * per[n]["launches_per_fwd"] * units(n) for n in per if n in fl)
    a_fl = sum(fl[n] * per[n]["launches_per_fwd"] for n in per if n in fl)
    mpp_fwd = w_fl / a_fl if a_fl > 0 else mpp
    return dict(bound="mfma", kernel=dom, achieved=per[dom]["tflops"], peak=peak, unit="TFLOP/s",
                frac=round(per[dom]["tflops"] / peak, 4), traffic=traffic, traffic_source=traffic_src,
                flops_per_launch=fl[dom], avg_launch_ms=per[dom]["avg_ms"],
                mfma_per_product=mpp, executed_mfma_frac=round(per[dom]["tflops"] * mpp / peak, 4),
                e2e_achieved=round(e2e, 1), e2e_peak=peak, e2e_frac=round(e2e / peak, 4),
                executed_flops_frac=round(executed / peak, 4),
                # matrix-pipe occupancy of the whole forward, NOT an efficiency figure: the FLOPs the pipes execute — every product of
                # this mode costs mfma_per_product MFMAs, i.e. x3 self-inflicted work in the default mode — over the f16 peak.  The
                # number that counts against the north star is e2e_frac (algorithmic FLOPs).
                mfma_units_whole_forward=round(mpp_fwd, 3),
                executed_mfma_whole_forward_frac=round(executed * mpp_fwd / peak, 4),
                # context, not a claim: what the matrix pipe sustains under the chip's power envelope on random operands (a committed probe result —
                # profiles/r04/mfma_power_probe.txt: nothing but MFMAs on registers; the nominal `peak` above is reached on all-zero operands only)
                envelope=dict(f16_mfma_random_operands_tflops=1614.0, fp8_scaled_mfma_random_operands_tflops=4292.0,
                              source="profiles/r04/mfma_power_probe.txt (scripts/probes/mfma_power_probe.hip)"),
                per_kernel=per)


def prob_err(a, b):
    return float(np.abs(1 / (1 + np.exp(-a.astype(np.float64))) - 1 / (1 + np.exp(-b.astype(np.float64)))).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default=os.environ.get("GLICLASS_DTYPE", "f32"), choices=["f32", "f16", "bf16"],
                    help="arithmetic mode of the headline number (default: the product's default, f32 = split-f16 MFMA products)")
    ap.add_argument("--throughput-dtype", default="f16", choices=["f16", "bf16", "none"], help="opt-in 16-bit mode measured beside the headline (N=1)")
    ap.add_argument("--config", default="base")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--labels", type=int, default=8)
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--dump-logits", default="", help=argparse.SUPPRESS)      # tests: rank 0 writes the step's [B, C] logits (.npy)
    ap.add_argument("--cpu-seqs", type=int, default=12, help="sequences timed on the CPU baseline, N=1 only (0 = skip); 12 = about 10 s on a 16-CPU share")
    ap.add_argument("--cpu-seqs-8", type=int, default=6, help="sequences of the OMP_NUM_THREADS=8 leg of the CPU baseline")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)        # CPU rehearsal of the multi-rank path (tests): gloo + StubRunner
    args = ap.parse_args()
    # BASELINE.json's configs as one flag each (explicit --batch / --seq / --scaling still win where given on the command line)
    alias = {"c2": ("small", 8, 512, "weak"), "c3": ("base", 64, 1024, "weak"), "c4": ("large", 256, 1024, "strong"), "c5": ("qwen-1.5b", 16, 2048, "weak")}
    if args.config in alias:
        cname, ab, asq, asc = alias[args.config]
        given = " ".join(sys.argv[1:])
        args.config = cname
        if "--batch" not in given: args.batch = ab
        if "--seq" not in given: args.seq = asq
        if args.scaling is None: args.scaling = asc
        # the CPU legs are a BOUNDED sample (about 10-30 s of CPU work): the larger configs cost ~10 s per sequence on the oracle
        if "--cpu-seqs" not in given and cname in ("large", "qwen-1.5b"): args.cpu_seqs, args.cpu_seqs_8 = 2, 1
    scaling_defaulted = args.scaling is None
    if args.scaling is None:
        args.scaling = "weak"

    # ---- the N ranks: from the launcher's environment, or started here ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.stub:
            import torch                      # device_count() does not initialise HIP on this image
            nd = torch.cuda.device_count()
            if nd < args.gpus:
                raise SystemExit(f"bench: --gpus {args.gpus} but only {nd} GPU(s) visible")
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}")

    # host threads: the CPUs this job may really use (affinity / cgroup quota, not the 256 a GPU box shows), shared between the
    # N ranks of one node, which generate their synthetic weights at the same time
    from gliclass.c_amd.hostinfo import effective_cpus
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(effective_cpus(), 64) // max(world, 1))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import torch
    dist = None
    if world > 1 or os.environ.get("GLC_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if args.stub:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            if torch.cuda.device_count() <= local_rank:
                raise SystemExit(f"bench: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
        world = dist.get_world_size()          # n_gpus is what really joined the group

    from gliclass.c_amd.config import CONFIGS
    cfg = CONFIGS[args.config]
    S, Cn = args.seq, args.labels
    if args.scaling == "strong":
        lo, hi = shard_rows(args.batch, world, rank)
        global_rows = args.batch
    else:
        lo, hi = 0, args.batch
        global_rows = args.batch * world
    B = hi - lo
    if B <= 0:
        raise SystemExit(f"bench: rank {rank} got no rows (--batch {args.batch} over {world} ranks)")

    W = hipl = modl = None
    if args.stub:
        runner = StubRunner(args, lo, B)
    else:
        from gliclass.c_amd import _lib, synth
        hipl, modl = _lib.hip(), _lib.model()
        if hipl.glc_device_count() <= local_rank:
            raise SystemExit("bench: no HIP device for this rank (the engine has no CPU path)")
        # weights: deterministic synthetic (no checkpoints offline), generated by the C host layer
        W = _lib.Weights()
        if modl.glc_weights_load(f"synthetic:{args.config}:42".encode(), C.byref(W)) != 0:
            raise SystemExit("bench: weight generation failed")
        if args.scaling == "strong":          # one global batch, this rank's contiguous rows of it
            gids, gmask, _ = synth.make_inputs(cfg, args.batch, S, Cn, seed=1234)
            ids, mask = np.ascontiguousarray(gids[lo:hi]), np.ascontiguousarray(gmask[lo:hi])
        else:
            ids, mask, _ = synth.make_inputs(cfg, B, S, Cn, seed=1234 + rank)
        runner = EngineRunner(args, cfg, W, local_rank, args.dtype, ids, mask)

    gathered = [None]

    def step():
        runner.step()
        if args.scaling == "strong" and dist is not None:
            runner.sync()                                  # the logits must be complete before RCCL reads them
            gathered[0] = gather_logits(dist, runner, world, rank, global_rows, Cn)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if not args.stub else dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.stub else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    own = []
    elapsed = time_steps(step, runner.sync, barrier, max_over_ranks, args.steps, args.warmup, own)
    seqs_per_s = global_rows * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    # every rank's own time per step (to its device sync, before the closing barrier): min / max over the ranks — a straggler is visible in the line
    own_ms = own[0] / args.steps * 1e3
    rank_ms = {"min": round(-max_over_ranks(-own_ms), 3), "max": round(max_over_ranks(own_ms), 3)}
    omp_threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    assert omp_threads >= 1, "host threads per rank"

    out = None
    if args.stub:
        if rank == 0:
            full = gathered[0] if gathered[0] is not None else runner.logits
            out = {"metric": "stub", "value": round(seqs_per_s, 2), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "scaling": args.scaling, "global_batch": global_rows, "rows_rank0": B, "step_calls_rank0": runner.calls,
                   "rank_ms_per_step": rank_ms, "omp_threads_per_rank": omp_threads,
                   "gathered_rows": [int(v) for v in (full[:, 0] / 16).tolist()] if args.scaling == "strong" else None}
    else:
        h = runner.h
        # HIP-event cross-check of the same loop on this rank's stream
        hipl.glc_timer_start(h)
        for _ in range(args.steps):
            runner.step()
        ev_ms = float(hipl.glc_timer_stop_ms(h)) / args.steps
        # per-step distribution (SURVEY.md §8d protocol: median + p10 / p90): every step timed on its own with HIP events on the engine's stream
        lap = []
        for _ in range(max(5, min(args.steps, 20))):
            hipl.glc_timer_start(h)
            runner.step()
            lap.append(float(hipl.glc_timer_stop_ms(h)))
        lap.sort()
        pct = lambda q: lap[min(len(lap) - 1, int(round(q * (len(lap) - 1))))]
        gather_ms = None
        if args.scaling == "strong" and dist is not None:        # the collective alone, HIP-event free (host clock, synchronous)
            runner.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                gather_logits(dist, runner, world, rank, global_rows, Cn)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - t0) / 20 * 1e3
        if rank == 0:
            logits = (gathered[0] if gathered[0] is not None else runner.logits).cpu().numpy()
            key = f"{args.config}:{args.batch}:{S}"
            roof = None
            mx_on = bool(hipl.glc_debug_last_forward_mx(h)) if args.dtype == "f32" else False
            mx_attn_on = bool(hipl.glc_debug_last_forward_mx_attention(h)) if mx_on else False
            if not args.no_profile:
                roof = profile_mode(hipl, h, runner.step, runner.sync, cfg, B, S, Cn, args.dtype, B / (ev_ms * 1e-3), key, mx_on, mx_attn_on)
            # boundary-inclusive step (SURVEY.md §8d protocol): host int64 ids/mask in, H2D, forward, D2H of the logits out
            host_ms = None
            if world == 1:
                hl = np.zeros((B, Cn), np.float32)
                c_out = C.c_int(0)
                hipl.glc_engine_forward(h, ids.ctypes.data, mask.ctypes.data, B, S, hl.ctypes.data, Cn, C.byref(c_out))
                t0 = time.perf_counter()
                for _ in range(max(2, min(args.steps, 5))):
                    hipl.glc_engine_forward(h, ids.ctypes.data, mask.ctypes.data, B, S, hl.ctypes.data, Cn, C.byref(c_out))
                host_ms = (time.perf_counter() - t0) / max(2, min(args.steps, 5)) * 1e3
            # ... and the same boundary as a serving loop would drive it: the NEXT batch's int64 ids / mask go host -> device on a copy
            # stream (pinned buffers, two device slots) while this batch's forward runs; every step ends with the logits on the host
            serve_ms = None
            if world == 1:
                try:
                    dev = f"cuda:{local_rank}"
                    pin_i, pin_m = torch.from_numpy(ids).pin_memory(), torch.from_numpy(mask).pin_memory()
                    slot = [(torch.empty_like(pin_i, device=dev), torch.empty_like(pin_m, device=dev)) for _ in range(2)]
                    evs = [torch.cuda.Event(), torch.cuda.Event()]
                    cs = torch.cuda.Stream(device=dev)
                    host_logits = torch.empty((B, Cn), dtype=torch.float32).pin_memory()

                    def upload(k):
                        with torch.cuda.stream(cs):
                            slot[k][0].copy_(pin_i, non_blocking=True); slot[k][1].copy_(pin_m, non_blocking=True)
                            evs[k].record(cs)

                    def serve(n):
                        upload(0)
                        t0 = time.perf_counter()
                        for i in range(n):
                            k = i & 1
                            evs[k].synchronize()
                            if hipl.glc_engine_forward_device(h, C.c_void_p(slot[k][0].data_ptr()), C.c_void_p(slot[k][1].data_ptr()), B, S, Cn, C.c_void_p(runner.logits.data_ptr())) != 0:
                                raise RuntimeError(hipl.glc_last_error().decode())
                            upload(k ^ 1)
                            runner.sync()
                            host_logits.copy_(runner.logits)
                        return (time.perf_counter() - t0) / n * 1e3

                    serve(2)
                    serve_ms = serve(max(4, min(args.steps, 10)))
                    if not np.array_equal(host_logits.numpy(), logits):
                        serve_ms = None      # (the loop must reproduce the timed forward's logits bit for bit, or its number is not reported)
                except Exception as e:       # plumbing only: never fail the line for it
                    print(f"bench: serving-loop leg skipped ({e})", file=sys.stderr)
            cpu = None
            ref_logits = rids = rmask = None
            if args.cpu_seqs > 0 and world == 1:
                print(f"bench: CPU baseline leg ({min(args.cpu_seqs, B)} + {args.cpu_seqs_8} sequences on the oracle) ...", file=sys.stderr, flush=True)
                cpu, ref_logits, rids, rmask = cpu_baseline(cfg, W.tensors, S, Cn, min(args.cpu_seqs, B), args.cpu_seqs_8)
                # parity check of the timed configuration itself: same seed => the oracle's rows are the first rows of the timed batch
                n = rids.shape[0]
                cpu["gpu_vs_cpu_max_prob_err"] = prob_err(logits[:n], ref_logits)
                cpu["rows_compared"] = n
            mode_txt = {"f32": ("f32 data; every product a_hi*w_hi in f16 MFMAs + both cross terms in one block-scaled fp8 MFMA (MX, ~2^-15 per product); projections: gemm256x on GX rows; attention: "
                                + ("attention_mx on MX tiles" if mx_attn_on else "split-f16 x3 MFMA products (GLC_MX_ATTN=0)") + " (the product's default mode)"
                                if mx_on else "f32 data, split-f16 x3 MFMA products (the default mode with GLICLASS_MX=0, or a forward too small for the 256-tile pipeline)"), "f16": "f16 MFMA operands (opt-in throughput mode)",
                        "bf16": "bf16 MFMA operands (opt-in throughput mode)"}[args.dtype]
            shape_txt = (f"gliclass-{args.config} (DeBERTa-v3 shape L={cfg.layers} H={cfg.hidden})" if cfg.backbone != 1 else
                         f"gliclass-{args.config} (decoder backbone L={cfg.layers} H={cfg.hidden}, {cfg.heads}q/{cfg.kv_heads}kv x {cfg.head_dim})")
            out = {
                "metric": f"sequences/sec at batch={args.batch} seq={S}, gliclass-{args.config}; %MFMA-peak",
                "value": round(seqs_per_s, 2), "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                "commit": os.environ.get("GLICLASS_BENCH_COMMIT") or git_head(),
                "dtype": args.dtype, "mode": mode_txt, "mx_projections": mx_on, "mx_attention": mx_attn_on, "data": "synthetic",
                "config": {"workload": f"{shape_txt}, batch={args.batch} seq={S} labels={Cn}, random-init weights (seed 42), full-length rows",
                           "global_batch": global_rows, "seq_len": S,
                           "timed": "device-resident int64 ids / mask in, device-resident logits out (glc_engine_forward_device); the boundary-inclusive steps are host_buffer_ms_per_step (synchronous H2D + forward + D2H) and serving_loop_ms_per_step (next batch's H2D under this forward)",
                           "parallelism": (f"batch-shard x{world}: one process per GPU, every rank a full batch, no data-path collective" if args.scaling == "weak" else
                                           f"batch-shard x{world}: one global batch split contiguously ({B} rows on rank 0), logits all-gathered to rank 0 every step (RCCL)")},
                "rank_ms_per_step": rank_ms, "omp_threads_per_rank": omp_threads,
                "hip_event_ms_per_step": round(ev_ms, 3),
                "step_ms_median": round(pct(0.5), 3), "step_ms_p10": round(pct(0.1), 3), "step_ms_p90": round(pct(0.9), 3), "step_ms_samples": len(lap),
                "finite": bool(np.isfinite(logits).all()),
            }
            if world > 1 and scaling_defaulted:
                out["config"]["parallelism"] += ("; BASELINE.json's c4 (one global batch of 256 split 32 per GPU, logits gathered) is "
                                                 "`bench.py --config c4 --gpus N` (= --config large --batch 256 --scaling strong)")
            if args.dump_logits:
                np.save(args.dump_logits, logits)
            if host_ms is not None:
                out["host_buffer_ms_per_step"] = round(host_ms, 3)     # H2D of ids/mask + forward + D2H of logits (glc_engine_forward)
            if serve_ms is not None:
                out["serving_loop_ms_per_step"] = round(serve_ms, 3)   # pinned host ids / mask of batch i + 1 uploaded on a copy stream under forward i; logits D2H every step
            if gather_ms is not None:
                out["gather_ms"] = round(gather_ms, 4)
            if roof:
                out["roofline"] = roof
            if cpu:
                out["cpu_baseline"] = cpu
                out["parity_ok"] = bool(cpu["gpu_vs_cpu_max_prob_err"] <= BAR)
                out["parity_bar"] = BAR
            else:
                out["parity_ok"] = None           # the oracle leg was skipped (--cpu-seqs 0 or N > 1): this line carries no parity evidence
            tdt = args.throughput_dtype
            if cpu and world == 1 and tdt != "none" and tdt != args.dtype:
                # The opt-in 16-bit throughput mode on the same workload: reported beside the headline number, never instead of
                # it — its operand rounding alone can exceed the 1e-3 bar on these random-weight models (docs/LOG_r01-r05.md §2).
                runner.close()
                r2 = EngineRunner(args, cfg, W, local_rank, tdt, ids, mask)
                for _ in range(2):
                    r2.step()
                r2.sync()
                nsteps = max(2, min(args.steps, 10))
                hipl.glc_timer_start(r2.h)
                for _ in range(nsteps):
                    r2.step()
                pms = float(hipl.glc_timer_stop_ms(r2.h)) / nsteps
                r2.sync()
                lg2 = r2.logits.cpu().numpy()
                pe = prob_err(lg2[: rids.shape[0]], ref_logits)
                tm = {"dtype": tdt, "value": round(B / (pms * 1e-3), 2), "unit": "sequences/s", "ms_per_step": round(pms, 3), "steps": nsteps,
                      "gpu_vs_cpu_max_prob_err": pe, "parity_ok": bool(pe <= BAR), "bar": BAR, "opt_in": f"GLICLASS_DTYPE={tdt}"}
                if not args.no_profile:
                    tm["roofline"] = profile_mode(hipl, r2.h, r2.step, r2.sync, cfg, B, S, Cn, tdt, B / (pms * 1e-3), key)
                out["throughput_mode"] = tm
                r2.close()
                runner = None
    if runner is not None:
        runner.close()
    if modl is not None:
        modl.glc_weights_free(C.byref(W))
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
