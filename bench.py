#!/usr/bin/env python3
"""Headline benchmark: samples/s of forward + log-det -> log p(x) (B, M) on synthetic CIFAR-10C-shaped
input, batch-sharded over N MI355X with one RCCL all-reduce of the summed log-likelihood per step.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over the GLOBAL batch (fixed as N grows: strong scaling, as
BASELINE.json's north_star asks).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_SAMPLE_STEP = {"cifar10": 2 * (2555904 + 65536), "mnist": None}   # MACs of one flow step x2
PEAK_F32_MFMA_TFLOPS = 157.3                                              # MI355X_MICROARCH.md, dense fp32 matrix
DIMS = {"cifar10": 3072, "mnist": 1024, "smap": 200, "atm": 38 * 144}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cifar10", choices=["cifar10", "mnist", "smap", "atm"])
    ap.add_argument("--global-batch", type=int, default=2097152)
    ap.add_argument("--chunk", type=int, default=262144, help="samples per kernel launch sequence on one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def synth(name, n, dev, seed):
    """Synthetic input resident in HBM: uint8-valued fp32 images (what Dequantization receives) or [0,1) series."""
    g = torch.Generator(device=dev).manual_seed(seed)
    C, H, W = {"cifar10": (3, 32, 32), "mnist": (1, 32, 32), "smap": (25, 8, 1), "atm": (38, 144, 1)}[name]
    out = torch.empty(n, C, H, W, device=dev, dtype=torch.float32)
    for i in range(0, n, 65536):                      # piecewise: randint materialises int64
        m = min(65536, n - i)
        if name in ("smap", "atm"):
            out[i:i + m] = torch.rand(m, C, H, W, device=dev, generator=g)
        else:
            out[i:i + m] = torch.randint(0, 256, (m, C, H, W), device=dev, generator=g).float()
    return out


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU
    box exposes every core of the host to os.cpu_count() but grants a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    else:
        n = min(n, 16)          # documented CPU share of a one-GPU box
    return n


def cpu_baseline(name, seconds):
    """The oracle (a CPU port of the reference path, pinned to the reference by tests/golden) timed on
    this box's host cores on a bounded sample of the same workload."""
    from oracle import flow_oracle as fo, params as op
    cores = host_cores()
    torch.set_num_threads(cores)
    ops, prior, M = fo.program(name)
    params = op.gen_params(op.param_spec(ops, prior, M), seed=0)
    B = 256
    g = torch.Generator().manual_seed(0)
    C, H, W = fo.CONFIGS[name][0]
    x = torch.rand(B, C, H, W, generator=g) if name in ("smap", "atm") else torch.randint(0, 256, (B, C, H, W), generator=g).float()
    u = torch.rand(B, C, H, W, generator=g)
    eps = [torch.randn(B, 1, H, W, generator=g)]
    with torch.no_grad():
        fo.flow_forward(ops, params, x, u, eps, init_actnorm=True)      # init + warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            fo.flow_forward(ops, params, x, u, eps)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 200:
                break
    return {"value": round(n * B / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d batches of %d %s-shaped samples, oracle/flow_oracle.py fp32, torch %d threads" % (n, B, name, cores)}


def main():
    a = parse()
    import contextflow_amd as cfa
    from contextflow_amd import dist as cdist
    from contextflow_amd.layers import _hip
    import torch.distributed as dist

    rank, local_rank, world = cdist.env_world()
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the HIP path has no CPU fallback)"
    # BENCH_REHEARSAL=1 (developer, one-GPU box): all ranks share device 0 and talk over gloo, to exercise the N > 1 code
    # path (sharding, broadcast, all-reduce, barriers) without N GPUs; the numbers of such a run mean nothing
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdist.init_process_group("gloo" if rehearsal else "nccl")
    _hip.lib()
    name = a.workload

    # model: reference init under a fixed seed; rank 0 runs the ActNorm data-dependent init, then broadcast
    torch.manual_seed(0)
    cfg, data_size, M = cfa.preset_config(name)
    model = cfa.create_model(cfg, data_size, M).to(dev)
    if rank == 0:
        with torch.no_grad():
            model(synth(name, 256, dev, seed=999))
    cdist.broadcast_parameters(model, src=0)

    G = a.global_batch
    lo, hi = cdist.shard_bounds(G, rank, world)
    x = synth(name, hi - lo, dev, seed=1000 + rank)
    nll_acc = torch.zeros(1, dtype=torch.float64, device=dev)
    events = []

    @torch.no_grad()                                   # density evaluation (experiment_cl.py:163-185 runs it under no_grad):
    def step(timed):                                   # no autograd tape, no W^-1 for the backward
        nll_acc.zero_()
        for c0 in range(0, hi - lo, a.chunk):
            xb = x[c0:c0 + a.chunk]
            _, logp = model(xb)
            _hip.call("cf_nll_sum", _hip.p(logp), _hip.p(nll_acc), logp.shape[0], logp.shape[1], _hip.stream())
        return cdist.allreduce_nll(nll_acc, hi - lo)           # RCCL all-reduce of [sum log p, count]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        red = step(False)
    model.step_events = events if FLOP_PER_SAMPLE_STEP.get(name) else None
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        red = step(True)
    fence()
    dt = time.perf_counter() - t0
    model.step_events = None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    bpd = cdist.mean_bits_per_dim(red.cpu(), DIMS[name])

    # roofline of the dominant kernel (k_flow_step, fp32 MFMA): algorithmic flops / measured duration
    roof = None
    if events:
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in events)
        flops = sum(b * FLOP_PER_SAMPLE_STEP[name] for _, _, b, _ in events)
        ach = flops / (ms * 1e-3) / 1e12
        per = {}
        for e0, e1, b, c in events:
            k = "C%d" % c
            per.setdefault(k, [0.0, 0.0])
            per[k][0] += e0.elapsed_time(e1); per[k][1] += b * FLOP_PER_SAMPLE_STEP[name]
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):          # PMC-measured HBM bytes (profiles/, tools/profile.sh), scaled to this run's launch size
            tj = json.load(open(tp))
            avg_b = sum(b for _, _, b, _ in events) / len(events)
            traffic = int(tj["k_flow_step_bytes_per_launch"] / tj["batch_per_launch"] * avg_b)
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                "kernel": "k_flow_step (Conv1x1+ActNorm+Coupling fused, v_mfma_f32_32x32x2_f32)",
                "launches": len(events), "avg_launch_ms": round(ms / len(events), 4),
                "per_level_tflops": {k: round(v[1] / (v[0] * 1e-3) / 1e12, 2) for k, v in per.items()},
                "kernel_time_share": round(ms * 1e-3 / dt, 3)}

    if rank == 0:
        label = {"cifar10": "CIFAR-10C conv flow", "mnist": "MNIST-R conv flow", "smap": "SMAP trans flow", "atm": "ATM trans flow"}[name]
        out = {"metric": "samples/s fwd+logdet (log p(x) (B,M)), %s" % label,
               "value": round(G * a.steps / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s --coupling %s, 3x32x32 uint8-valued fp32 input, generalist, reference init" % (name, cfg["coupling"])
                          if name == "cifar10" else "%s --coupling %s" % (name, cfg["coupling"]),
                          "global_batch": G, "per_gpu_batch": hi - lo, "chunk": a.chunk, "parallelism": "dp%d" % world,
                          "collective": "all_reduce(sum log p, count) fp64 x2 per step"},
               "bits_per_dim": round(bpd, 6), "roofline": roof}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name, a.cpu_seconds)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
