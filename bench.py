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
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before the first GPU call: RCCL needs dmabuf IPC on this pool

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3                                              # MI355X_MICROARCH.md, dense fp32 matrix
DIMS = {"cifar10": 3072, "mnist": 1024, "smap": 200, "atm": 38 * 144}
SHAPES = {"cifar10": (3, 32, 32), "mnist": (1, 32, 32), "smap": (25, 8, 1), "atm": (38, 144, 1)}
LABEL = {"cifar10": "CIFAR-10C conv flow", "mnist": "MNIST-R conv flow", "smap": "SMAP trans flow", "atm": "ATM trans flow"}
# Algorithmic flop of ONE launch of the dominant kernel per sample (counted on the reference with hooks):
#  conv flows: k_flow_step = Conv1x1 (C^2 HW MAC) + coupling net (C/2*2C + 9*2C*2C + 2C*C = 39 C^2 MAC per pixel)
#              = 2 * 40 C^2 HW flop  (cifar10: 5 242 880 at every level; mnist: 1 310 720 at C = 8, 5 242 880 at C = 32);
#  smap: k_vit_step = Conv1x1 26x26x8 + SimpleViT linears 454 688 + attention QK^T / PV 12 288 MAC = 944 768 flop.
VIT_FLOP_PER_SAMPLE = {"smap": 2 * (26 * 26 * 8 + 454688 + 12288)}


def step_flop(C, HW):
    return 80 * C * C * HW


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cifar10", choices=["cifar10", "mnist", "smap", "atm"])
    ap.add_argument("--global-batch", type=int, default=2097152)
    ap.add_argument("--chunk", type=int, default=262144, help="samples per kernel launch sequence on one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the mnist / smap / small-batch lines (N = 1 only)")
    return ap.parse_args()


def synth(name, n, dev, seed):
    """Synthetic input resident in HBM: uint8-valued fp32 images (what Dequantization receives) or [0,1) series."""
    g = torch.Generator(device=dev).manual_seed(seed)
    C, H, W = SHAPES[name]
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


def cpu_baseline(name, B=256, iters=10, warmups=3):
    """The oracle (a CPU port of the reference path, pinned to the reference by tests/golden) timed on this box's host
    cores on a bounded sample of the same workload, as BASELINE.md section 3 prescribes: torch threads = usable cores,
    no_grad, 3 warm-ups, median of >= 10 iterations of one batch."""
    from oracle import flow_oracle as fo, params as op
    cores = host_cores()
    torch.set_num_threads(cores)
    ops, prior, M = fo.program(name)
    params = op.gen_params(op.param_spec(ops, prior, M), seed=0)
    g = torch.Generator().manual_seed(0)
    C, H, W = fo.CONFIGS[name][0]
    x = torch.rand(B, C, H, W, generator=g) if name in ("smap", "atm") else torch.randint(0, 256, (B, C, H, W), generator=g).float()
    u = torch.rand(B, C, H, W, generator=g)
    eps = [torch.randn(B, 1, H, W, generator=g)]
    times = []
    with torch.no_grad():
        fo.flow_forward(ops, params, x, u, eps, init_actnorm=True)      # ActNorm init
        for i in range(warmups + iters):
            t0 = time.perf_counter()
            fo.flow_forward(ops, params, x, u, eps)
            if i >= warmups:
                times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B / med, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "median of %d forward passes over one batch of %d %s-shaped samples after %d warm-ups, "
                      "oracle/flow_oracle.py fp32, torch %d threads" % (iters, B, name, warmups, cores)}


def build(name, dev, rank=0):
    """Reference init under a fixed seed; rank 0 runs the ActNorm data-dependent init on 256 synthetic samples."""
    import contextflow_amd as cfa
    torch.manual_seed(0)
    cfg, data_size, M = cfa.preset_config(name)
    model = cfa.create_model(cfg, data_size, M).to(dev)
    if rank == 0:
        with torch.no_grad():
            model(synth(name, 256, dev, seed=999))
    return model, cfg


def kernel_events(model, name):
    """Turn on the HIP-event probes around the dominant kernel of this workload; returns the list they fill."""
    from contextflow_amd.layers import coupling
    ev = []
    if name in VIT_FLOP_PER_SAMPLE:
        coupling.VIT_EVENTS = ev
    else:
        model.step_events = ev
    return ev


def roofline(events, name, dt):
    """Roofline of the dominant kernel (fp32 MFMA): algorithmic flop of the launches / their measured durations."""
    if not events:
        return None
    vit = name in VIT_FLOP_PER_SAMPLE
    ms = sum(e[0].elapsed_time(e[1]) for e in events)
    flop = sum(e[2] * (VIT_FLOP_PER_SAMPLE[name] if vit else step_flop(e[3], e[4])) for e in events)
    ach = flop / (ms * 1e-3) / 1e12
    # multiply-adds actually issued to the matrix pipe: the Winograd F(2x2,3x3) form of the 3x3 executes 32 of its 72 C^2 HW
    # (cf_flow_step_fwd: 16x16 images always, 8x8 / 4x4 from 1024 / 2048 samples per launch)
    wino = lambda e: (not vit) and os.environ.get("CONTEXTFLOW_DIRECT_CONV") != "1" and (
        (e[3] in (8, 16) and e[4] == 256) or (e[3] == 32 and e[2] >= 1024) or (e[3] == 64 and e[2] >= 2048))
    flop_exec = sum(e[2] * (VIT_FLOP_PER_SAMPLE[name] if vit else step_flop(e[3], e[4]) * (0.5 if wino(e) else 1.0)) for e in events)
    ach_exec = flop_exec / (ms * 1e-3) / 1e12
    per = {}
    for e in events:
        k = "vit" if vit else "C%d" % e[3]
        per.setdefault(k, [0.0, 0.0])
        per[k][0] += e[0].elapsed_time(e[1])
        per[k][1] += e[2] * (VIT_FLOP_PER_SAMPLE[name] if vit else step_flop(e[3], e[4]))
    traffic = None
    tp = os.path.join(ROOT, "profiles", {"cifar10": "traffic.json", "mnist": "r2_mnist3_traffic.json",
                                         "smap": "r2_smap1_traffic.json"}.get(name, "none"))
    if os.path.exists(tp):          # PMC-measured HBM bytes of the dominant kernel (profiles/, tools/profile.sh), scaled to this run's launch size
        tj = json.load(open(tp))
        avg_b = sum(e[2] for e in events) / len(events)
        traffic = int(tj["k_flow_step_bytes_per_launch"] / tj["batch_per_launch"] * avg_b)
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
            # `achieved` counts the ALGORITHMIC flop of the reference's direct convolutions; `executed` is what
            # the matrix pipe really ran (Winograd form of the 3x3: half of it) - frac > 1 is the algorithm, not the hardware
            "executed": round(ach_exec, 2), "executed_frac": round(ach_exec / PEAK_F32_MFMA_TFLOPS, 4),
            "algorithm": "direct" if abs(ach_exec - ach) < 1e-9 else "Winograd F(2x2,3x3) for the 3x3 of the coupling nets (fp32, 40 of 80 C^2 HW multiply-adds per sample-step executed)",
            "kernel": "k_vit_step (Conv1x1+ActNorm+TransCoupling fused, v_mfma_f32_32x32x2_f32)" if vit else
                      "k_flow_step / k_flow_step_small (Conv1x1+ActNorm+Coupling fused; v_mfma_f32_32x32x2_f32 / 16x16x4_f32, the 3x3 in Winograd F(2x2,3x3) form on 16x16x4 tiles)",
            "launches": len(events), "avg_launch_ms": round(ms / len(events), 4),
            "per_level_tflops": {k: round(v[1] / (v[0] * 1e-3) / 1e12, 2) for k, v in per.items()},
            "kernel_time_share": round(ms * 1e-3 / dt, 3)}


def total_flop_per_sample(name):
    return {"cifar10": 62914560, "mnist": 13107200, "smap": 7361536 + 2 * 98304}[name]


def secondary_throughput(name, dev, G, chunk, steps, warmup, cpu):
    """The other single-GPU BASELINE.json configs at a saturating batch: same loop as the headline, one GPU."""
    from contextflow_amd.layers import _hip, coupling
    model, cfg = build(name, dev)
    x = synth(name, G, dev, seed=2000)
    nll = torch.zeros(1, dtype=torch.float64, device=dev)

    @torch.no_grad()
    def step():
        nll.zero_()
        for c0 in range(0, G, chunk):
            _, logp = model(x[c0:c0 + chunk])
            _hip.call("cf_nll_sum", _hip.p(logp), _hip.p(nll), logp.shape[0], logp.shape[1], _hip.stream())
    for _ in range(warmup):
        step()
    ev = kernel_events(model, name)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.step_events, coupling.VIT_EVENTS = None, None
    out = {"metric": "samples/s fwd+logdet (log p(x) (B,M)), %s" % LABEL[name], "value": round(G * steps / dt, 1),
           "unit": "samples/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
           "config": {"workload": "%s --coupling %s, generalist, reference init" % (name, cfg["coupling"]), "global_batch": G, "chunk": chunk},
           "bits_per_dim": round(float(-(nll / G) / (DIMS[name] * math.log(2.0))), 6), "roofline": roofline(ev, name, dt)}
    if cpu:
        out["cpu_baseline"] = cpu_baseline(name)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    del model, x
    torch.cuda.empty_cache()
    return out


def secondary_training(name, dev, B, iters, graph):
    """SURVEY.md 8(f)1, the caller right after the path (experiment_cl.py:123-136): one optimisation step = forward,
    loss (cross-entropy over the M class mixtures of logp / D), hand-written backward, AdamW - as ONE captured HIP graph (FlowSequential.capture_train_step) at a saturating
    batch and at the reference's batch of 256, and eagerly launched (the reference's loop as written; host-bound when the
    box's cores are busy)."""
    model, cfg = build(name, dev)
    x = synth(name, B, dev, seed=4000)
    M = 10
    gt = torch.randint(0, M, (B,), device=dev)
    inv = 1.0 / DIMS[name]
    loss_fn = lambda logp, y: torch.nn.functional.cross_entropy(logp * inv, y)
    # the reference's optimizer (model.py:289: AdamW(params, lr)); under capture its fused implementation: ONE multi-tensor
    # kernel per step (the foreach form falls back to two launches per parameter on the 0-dim step tensors of capturable mode)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True) if graph and os.environ.get("CF_BENCH_FOREACH_ADAMW") != "1" \
        else torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=graph)
    if graph:
        step = model.capture_train_step(x, loss_fn, opt)
        run = lambda: step(x, gt)
    else:
        def run():
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(model.log_prob(x), gt)
            loss.backward()
            opt.step()
            return loss
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # algorithmic dense work of a step: forward + data gradients + weight gradients = 3x the forward contractions
    tf = 3 * total_flop_per_sample(name) * B / dt / 1e12
    out = {"metric": "samples/s training step (fwd + bwd + AdamW), %s" % LABEL[name], "value": round(B / dt, 1), "unit": "samples/s",
           "config": {"workload": "%s --coupling %s" % (name, cfg["coupling"]), "batch": B, "iters": iters,
                      "launch": "one captured HIP graph" if graph else "eager"},
           "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss), 5),
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                        "basis": "whole step, 3x the forward's dense flop (forward, data gradients, weight gradients)"}}
    del model, x, opt
    torch.cuda.empty_cache()
    return out


def secondary_small_batch(name, dev, B, cpu, iters=300):
    """The reference's operating point (config.py:10: batch 256; BASELINE config 1: 64): latency of ONE eval forward
    through the public API (`flow.log_prob(x)` under no_grad), eagerly launched and - the default once a shape repeats -
    replayed from the captured HIP graph.  roofline = all dense flop of the call / its wall time (whole call, not one
    kernel: at these sizes the call is launch- and latency-bound)."""
    model, cfg = build(name, dev)
    x = synth(name, B, dev, seed=3000)
    res = {}
    with torch.no_grad():
        for mode in ("eager", "graph"):
            model.auto_graph = mode == "graph"
            for _ in range(20):
                model.log_prob(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                model.log_prob(x)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / iters
    best = min(res.values())
    tf = total_flop_per_sample(name) * B / best / 1e12
    out = {"metric": "samples/s fwd+logdet, %s, one batch of %d per call" % (LABEL[name], B), "value": round(B / best, 1),
           "unit": "samples/s", "config": {"workload": "%s --coupling %s" % (name, cfg["coupling"]), "batch": B, "iters": iters},
           "us_per_call": {k: round(v * 1e6, 1) for k, v in res.items()},
           "roofline": {"bound": "mfma", "achieved": round(tf, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None, "basis": "whole call (launch-bound regime)"}}
    if cpu:
        out["cpu_baseline"] = cpu_baseline(name, B=B)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    del model
    torch.cuda.empty_cache()
    return out


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

    # model: reference init under a fixed seed; rank 0 runs the ActNorm data-dependent init, then ONE flat broadcast
    model, cfg = build(name, dev, rank)
    cdist.broadcast_parameters(model, src=0)

    G = a.global_batch
    lo, hi = cdist.shard_bounds(G, rank, world)
    x = synth(name, hi - lo, dev, seed=1000 + rank)
    nll_acc = torch.zeros(1, dtype=torch.float64, device=dev)
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
    events = kernel_events(model, name)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        red = step(True)
    fence()
    dt = time.perf_counter() - t0
    from contextflow_amd.layers import coupling as _cpl
    model.step_events, _cpl.VIT_EVENTS = None, None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    bpd = cdist.mean_bits_per_dim(red.cpu(), DIMS[name])
    roof = roofline(events, name, dt)
    comm = {"backend": dist.get_backend() if world > 1 else None, "world_size": world,
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
            "rehearsal_gloo_on_one_device": rehearsal}

    if rank == 0:
        out = {"metric": "samples/s fwd+logdet (log p(x) (B,M)), %s" % LABEL[name],
               "value": round(G * a.steps / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s --coupling %s, 3x32x32 uint8-valued fp32 input, generalist, reference init" % (name, cfg["coupling"])
                          if name == "cifar10" else "%s --coupling %s" % (name, cfg["coupling"]),
                          "global_batch": G, "per_gpu_batch": hi - lo, "chunk": a.chunk, "parallelism": "dp%d" % world,
                          "collective": "all_reduce(sum log p, count) fp64 x2 per step"},
               "bits_per_dim": round(bpd, 6), "roofline": roof, "comm": comm}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        if world == 1 and not a.no_secondary and name == "cifar10":
            # the other single-GPU configs of BASELINE.json and the batch sizes of SURVEY.md 8(d); the headline fields above
            # are unaffected (measured first, their buffers released)
            del model, x
            torch.cuda.empty_cache()
            cpu = not a.no_cpu_baseline
            out["secondary"] = [
                secondary_throughput("mnist", dev, 2097152, 262144, 3, 1, cpu),
                secondary_throughput("smap", dev, 4194304, 524288, 3, 1, cpu),
                secondary_small_batch("cifar10", dev, 64, cpu),
                secondary_small_batch("cifar10", dev, 256, cpu=False),     # its CPU baseline is the headline's (batch 256)
                secondary_small_batch("mnist", dev, 64, cpu),
                secondary_small_batch("smap", dev, 256, cpu=False),        # BASELINE config 4 at the reference's default batch
                secondary_training("cifar10", dev, 16384, 10, graph=True),
                secondary_training("cifar10", dev, 16384, 10, graph=False),
                secondary_training("cifar10", dev, 256, 50, graph=True),
            ]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
