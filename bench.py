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
#  smap: k_vit_step = Conv1x1 26x26x8 + SimpleViT linears 454 688 + attention QK^T / PV 12 288 MAC = 944 768 flop (the
#        library reports it, and what its MFMAs execute: cf_vit_step_macs).
VIT_STEP = {"smap": (26, 6)}                                              # (C, depth) of the one-kernel transformer step
TRAFFIC_JSON = {"cifar10": "r4_prof4_traffic.json", "mnist": "r4_mnist2_traffic.json", "smap": "r4_smap2_traffic.json"}


def vit_flop_per_sample(name, what):
    from contextflow_amd.layers import _hip
    return 2 * _hip.lib().cf_vit_step_macs(VIT_STEP[name][0], VIT_STEP[name][1], what)

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
    ap.add_argument("--train", action="store_true", help="time the data-parallel TRAINING step (SURVEY 8(f)1) instead of the forward: "
                    "one captured HIP graph per rank with the gradient all-reduces inside, weak scaling")
    ap.add_argument("--train-batch", type=int, default=16384, help="samples per rank and training step (--train)")
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
    if name in VIT_STEP:
        coupling.VIT_EVENTS = ev
    else:
        model.step_events = ev
    return ev


def roofline(events, name, dt):
    """Roofline of the dominant kernel (fp32 MFMA) from the HIP events around its launches.  `achieved` / `frac` count the
    USEFUL multiply-adds the matrix pipe executed, so frac <= 1 by construction: for the conv step kernels what the library
    reports per launch (cf_flow_step_macs - the Winograd F(2x2,3x3) form of the 3x3 runs 16 of its 36 C^2 HW; those kernels
    have no padding rows), for the transformer step cf_vit_step_macs(.., 2) - its fused 52 x 52 products WITHOUT the padding
    rows of the 32-row tiles (the padding-inclusive rate is `executed_incl_padding_tflops`).  `algorithmic_tflops` is the
    reference's flop count over the same time, `algorithmic_frac` that over the peak (it may exceed 1: both
    kernels run fewer multiplications than the reference's formulation) and `algorithmic_speedup` = algorithmic / useful."""
    if not events:
        return None
    from contextflow_amd.layers import _hip
    L = _hip.lib()
    vit = name in VIT_STEP
    ms = sum(e[0].elapsed_time(e[1]) for e in events)
    hw = lambda e: int(round(math.sqrt(e[4])))
    alg = lambda e: e[2] * (vit_flop_per_sample(name, 0) if vit else step_flop(e[3], e[4]))
    exe = lambda e: e[2] * (vit_flop_per_sample(name, 2) if vit else 2 * L.cf_flow_step_macs(e[2], e[3], hw(e), hw(e), 0))
    pad = lambda e: e[2] * (vit_flop_per_sample(name, 1) if vit else 2 * L.cf_flow_step_macs(e[2], e[3], hw(e), hw(e), 0))
    flop, flop_exec, flop_pad = sum(alg(e) for e in events), sum(exe(e) for e in events), sum(pad(e) for e in events)
    ach, ach_exec, ach_pad = (f / (ms * 1e-3) / 1e12 for f in (flop, flop_exec, flop_pad))
    per = {}
    for e in events:
        k = "vit" if vit else "C%d" % e[3]
        per.setdefault(k, [0.0, 0.0, 0.0])
        per[k][0] += e[0].elapsed_time(e[1])
        per[k][1] += alg(e)
        per[k][2] += exe(e)
    traffic = None
    tp = os.path.join(ROOT, "profiles", TRAFFIC_JSON.get(name, "none"))
    if os.path.exists(tp):          # PMC-measured HBM bytes of the dominant kernel (profiles/, tools/profile.sh), scaled to this run's launch size
        tj = json.load(open(tp))
        avg_b = sum(e[2] for e in events) / len(events)
        traffic = int(tj["k_flow_step_bytes_per_launch"] / tj["batch_per_launch"] * avg_b)
    out = {"bound": "mfma", "achieved": round(ach_exec, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": round(ach_exec / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
           "basis": "useful multiply-adds executed by the matrix pipe (x2) / HIP-event time of the launches",
           "algorithmic_tflops": round(ach, 2), "algorithmic_frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
           "algorithmic_speedup": round(ach / ach_exec, 3),
           "algorithm": ("q.k through Wq^T Wk and value / output through Wout Wv (two 52 x 52 products per attention block instead of "
                         "52 -> 192 and 64 -> 52; fp64-folded once per parameter version), LayerNorm affine folded into the next Linear") if vit else
                        ("as the reference computes it" if abs(ach_exec - ach) < 1e-9 * ach else
                         "Winograd F(2x2,3x3) for the 3x3 of the coupling nets where dispatched (fp32; 20 of 40 C^2 HW multiply-adds per sample-step executed)"),
           "kernel": "k_vit_step (Conv1x1+ActNorm+TransCoupling fused, v_mfma_f32_32x32x2_f32)" if vit else
                     "k_flow_step / k_flow_step_small (Conv1x1+ActNorm+Coupling fused; v_mfma_f32_32x32x2_f32 / 16x16x4_f32, the 3x3 in Winograd F(2x2,3x3) form on 16x16x4 tiles)",
           "launches": len(events), "avg_launch_ms": round(ms / len(events), 4),
           "per_level_executed_tflops": {k: round(v[2] / (v[0] * 1e-3) / 1e12, 2) for k, v in per.items()},
           "per_level_algorithmic_tflops": {k: round(v[1] / (v[0] * 1e-3) / 1e12, 2) for k, v in per.items()},
           "kernel_time_share": round(ms * 1e-3 / dt, 3)}
    if vit:
        out["executed_incl_padding_tflops"] = round(ach_pad, 2)
    return out


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


def reference_loss(name):
    """The reference's training loss on logp = dim_inv * log_prob (NaN scrubbed first: `logp[logp != logp] = 0`, written
    as a select so that it can be captured in a graph).  Classification flows (experiment_cl.py:127-133, alpha = 1e-3):
    CE(logp, gt) + alpha * mean(-logsigmoid(logsumexp_m logp)); anomaly detection without labels (experiment_ad.py:61,
    204-210, alpha = 1e2): alpha * mean(-logsigmoid(logp))."""
    inv = 1.0 / DIMS[name]
    F = torch.nn.functional

    def scrub(logp):
        logp = logp * inv
        return torch.where(logp != logp, torch.zeros_like(logp), logp)
    if name in ("smap", "atm"):
        return lambda logp, y: 1e2 * (-F.logsigmoid(scrub(logp))).mean()

    def cl(logp, y):
        logp = scrub(logp)
        return F.cross_entropy(logp, y) + 1e-3 * (-F.logsigmoid(torch.logsumexp(logp, -1))).mean()
    return cl


def train_flop_per_sample(name, B):
    """(algorithmic, executed) dense flop per sample of one training step.  Conv flows: per flow step forward + data
    gradients + weight gradients = 3 x 80 C^2 HW algorithmic; executed as the library reports for the kernels dispatched
    at this batch size (Winograd form in the taping forward and in the 3x3 weight gradient, direct transposed 3x3 in the
    data-gradient chain).  Transformer flow: 3 x the forward's Linear / attention flop (the backward re-runs the
    conditioner on top of that; not counted)."""
    from contextflow_amd.layers import _hip
    L = _hip.lib()
    if name in VIT_STEP:
        f = 3 * total_flop_per_sample(name)
        return f, f
    levels = {"cifar10": [(16, 16, 4), (32, 8, 4), (64, 4, 4)], "mnist": [(8, 16, 2), (32, 8, 2)]}[name]
    alg = sum(3 * step_flop(C, H * H) * n for C, H, n in levels)
    exe = sum(2 * n * (L.cf_flow_step_macs(B, C, H, H, 1) + L.cf_flow_step_macs(B, C, H, H, 2) + L.cf_step_wgrads_macs(B, C, H, H))
              for C, H, n in levels)
    return alg, exe


def cpu_train_baseline(name, B=256, iters=5, warmups=1):
    """The oracle's forward under torch.autograd + AdamW on the host cores: one training step of the same loss on one
    batch (`kind: "port"`), median of `iters`."""
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
    gt = torch.randint(0, M, (B,), generator=g)
    with torch.no_grad():
        fo.flow_forward(ops, params, x, u, eps, init_actnorm=True)
    leaves = [v.requires_grad_(True) for v in params.values() if v.is_floating_point()]
    opt = torch.optim.AdamW(leaves, lr=1e-4)
    loss_fn = reference_loss(name)
    times = []
    for i in range(warmups + iters):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss_fn(fo.flow_forward(ops, params, x, u, eps)[1], gt).backward()
        opt.step()
        if i >= warmups:
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B / med, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "median of %d training steps (oracle forward under torch.autograd, the reference's loss, AdamW) on one batch of "
                      "%d %s-shaped samples after %d warm-up, torch %d threads" % (iters, B, name, warmups, cores)}


def secondary_training(name, dev, B, iters, graph, cpu=False, own_adamw=False):
    """SURVEY.md 8(f)1, the caller right after the path (experiment_cl.py:123-136 / experiment_ad.py:199-213): one
    optimisation step = forward, the reference's loss, hand-written backward, AdamW - as ONE captured HIP graph
    (FlowSequential.capture_train_step) at a saturating batch and at the reference's batch of 256, and eagerly launched
    (the reference's loop as written; host-bound when the box's cores are busy)."""
    model, cfg = build(name, dev)
    x = synth(name, B, dev, seed=4000)
    M = model.mixtures
    gt = torch.randint(0, M, (B,), device=dev)
    loss_fn = reference_loss(name)
    # the reference's optimizer (model.py:289: AdamW(params, lr)); under capture its fused implementation: ONE multi-tensor
    # kernel per step (the foreach form falls back to two launches per parameter on the 0-dim step tensors of capturable mode)
    if own_adamw:
        # the same update through contextflow_amd.optim.FusedAdamW: the whole parameter table in one or two launches (torch's fused
        # multi-tensor kernel: 4 launches of 43 us for cifar10's 135 tensors, 16 of 15 us for smap's 571)
        import contextflow_amd as cfa
        opt = cfa.optim.FusedAdamW(model.parameters(), lr=1e-4)
    elif graph and os.environ.get("CF_BENCH_FOREACH_ADAMW") != "1":
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=graph)
    if graph:
        step = model.capture_train_step(x, loss_fn, opt, data_parallel=False)     # single-GPU line; `bench.py --train` is the data-parallel step
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
    alg, exe = train_flop_per_sample(name, B)
    tf_alg, tf_exe = alg * B / dt / 1e12, exe * B / dt / 1e12
    out = {"metric": "samples/s training step (fwd + bwd + AdamW), %s" % LABEL[name], "value": round(B / dt, 1), "unit": "samples/s",
           "config": {"workload": "%s --coupling %s" % (name, cfg["coupling"]), "batch": B, "iters": iters,
                      "launch": "one captured HIP graph" if graph else "eager",
                      "optimizer": "contextflow_amd.optim.FusedAdamW" if own_adamw else "torch.optim.AdamW(fused=True)" if graph else "torch.optim.AdamW",
                      "loss": "experiment_ad.py:207 (1e2 * -logsigmoid(logp / D), NaN scrubbed)" if M == 1 else
                              "experiment_cl.py:128-133 (NaN scrub, CE + 1e-3 * -logsigmoid(logsumexp))"},
           "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss), 5),
           "roofline": {"bound": "mfma", "achieved": round(tf_exe, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf_exe / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                        "algorithmic_tflops": round(tf_alg, 2), "algorithmic_speedup": round(tf_alg / tf_exe, 3),
                        "basis": ("whole step, ALGORITHMIC: 3 x the reference's forward flop (forward, data gradients, weight gradients; the conditioner "
                                  "re-run of the backward kernel is not counted) / wall time") if name in VIT_STEP else
                                 "whole step: dense multiply-adds executed by the matrix pipe (forward, data gradients, weight gradients) / wall time"}}
    if cpu:
        out["cpu_baseline"] = cpu_train_baseline(name)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    del model, x, opt
    torch.cuda.empty_cache()
    return out


def cpu_sampling_baseline(name, B=256, iters=5, warmups=1):
    """The oracle's sampling direction on the host cores (flowsequential.py:32-39: prior draw, every layer's reverse from last
    to first; mnist topology - the only one whose `sample` runs in the reference)."""
    from oracle import flow_oracle as fo, params as op
    cores = host_cores()
    torch.set_num_threads(cores)
    ops, prior, M = fo.program(name)
    params = op.gen_params(op.param_spec(ops, prior, M), seed=0)
    g = torch.Generator().manual_seed(0)
    C, H, W = fo.CONFIGS[name][0]
    x = torch.randint(0, 256, (B, C, H, W), generator=g).float()
    aug = [i for i, o in enumerate(ops) if o[0] == "augment"][0]
    times = []
    with torch.no_grad():
        fo.flow_forward(ops, params, x, torch.rand(B, C, H, W, generator=g), [torch.randn(B, 1, H, W, generator=g)], init_actnorm=True)
        mG, sG, wG = params["dist.mG"], params["dist.sG"], params["dist.wG"]
        for i in range(warmups + iters):
            t0 = time.perf_counter()
            k = torch.multinomial(torch.softmax(wG[1], -1), B, replacement=True, generator=g)              # gaussian.py:163-169
            z = mG[1][k] + torch.nn.functional.softplus(sG[1][k]) * torch.randn(B, *mG.shape[2:], generator=g)
            h = fo.flow_inverse_layers(ops[aug + 1:], params, z)
            h = fo.flow_inverse_layers(ops[:aug], params, h[:, :C])                                         # Augment.reverse drops the noise channel
            if i >= warmups:
                times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B / med, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "median of %d sampling passes (oracle/flow_oracle.py: prior draw + reverse chain) of %d %s-shaped samples after %d warm-up, "
                      "torch %d threads" % (iters, B, name, warmups, cores)}


def secondary_sampling(name, dev, B, iters, cpu):
    """north_star: "forward+inverse".  `flow.sample(B)` (flowsequential.py:32-39): mixture draw, then Coupling^-1, ActNorm^-1,
    Conv1x1^-1 of every step as ONE kernel (k_flow_step_inv), Squeeze^-1, Augment / pre-processing reverses."""
    from contextflow_amd.layers import _hip
    L = _hip.lib()
    model, cfg = build(name, dev)
    with torch.no_grad():
        for _ in range(2):
            model.sample(B)
        model.inv_events = ev = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            smp = model.sample(B)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    model.inv_events = None
    ms = sum(e[0].elapsed_time(e[1]) for e in ev)
    hw = lambda e: int(round(math.sqrt(e[4])))
    exe = sum(2 * e[2] * L.cf_flow_step_macs(e[2], e[3], hw(e), hw(e), 3) for e in ev)
    alg = sum(e[2] * step_flop(e[3], e[4]) for e in ev)
    per = {}
    for e in ev:
        v = per.setdefault("C%d" % e[3], [0.0, 0.0])
        v[0] += e[0].elapsed_time(e[1])
        v[1] += 2 * e[2] * L.cf_flow_step_macs(e[2], e[3], hw(e), hw(e), 3)
    out = {"metric": "samples/s sampling (prior draw + inverse flow), %s" % LABEL[name], "value": round(B / dt, 1), "unit": "samples/s",
           "config": {"workload": "%s --coupling %s, flow.sample" % (name, cfg["coupling"]), "batch": B, "iters": iters},
           "ms_per_step": round(dt * 1e3, 3), "finite": bool(torch.isfinite(smp).all()),
           "roofline": {"bound": "mfma", "achieved": round(exe / (ms * 1e-3) / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(exe / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                        "algorithmic_tflops": round(alg / (ms * 1e-3) / 1e12, 2),
                        "basis": "multiply-adds executed by the matrix pipe (x2) / HIP-event time of the k_flow_step_inv launches",
                        "kernel": "k_flow_step_inv (Coupling^-1, ActNorm^-1, Conv1x1^-1 fused; conditioner as in the forward)",
                        "launches": len(ev), "avg_launch_ms": round(ms / max(len(ev), 1), 4),
                        "per_level_executed_tflops": {k: round(v[1] / (v[0] * 1e-3) / 1e12, 2) for k, v in per.items()},
                        "kernel_time_share": round(ms * 1e-3 / (dt * iters), 3)}}
    if cpu:
        out["cpu_baseline"] = cpu_sampling_baseline(name)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    del model
    torch.cuda.empty_cache()
    return out


SPECIALIST_CTX = {"cifar10": dict(contexts=[15, 5], enc_emb="onehot", enc_type="uniform", contextflow=True)}


def cpu_specialist_baseline(name, B=256, iters=5, warmups=1):
    """The oracle's specialist forward (context encoders, per-sample Conv1x1 / ActNorm, CN(c) branches, context-shifted
    mixtures) on the host cores, one batch."""
    from oracle import flow_oracle as fo, params as op
    cores = host_cores()
    torch.set_num_threads(cores)
    ctx = SPECIALIST_CTX[name]
    ops, prior, M = fo.program(name)
    params = op.gen_params(op.param_spec(ops, prior, M, ctx), seed=0)
    g = torch.Generator().manual_seed(0)
    C, H, W = fo.CONFIGS[name][0]
    x = torch.randint(0, 256, (B, C, H, W), generator=g).float()
    u = torch.rand(B, C, H, W, generator=g)
    eps = [torch.randn(B, 1, H, W, generator=g)]
    context = torch.stack([torch.randint(0, k, (B,), generator=g) for k in ctx["contexts"]], 1)
    n_enc = sum(1 for o in ops if o[0] in ("conv1x1", "actnorm", "coupling", "transcoupling"))
    cn = lambda: [torch.rand(B, fo.ctx_width(ctx), generator=g) for _ in range(n_enc)]
    times = []
    with torch.no_grad():
        fo.flow_forward(ops, params, x, u, eps, init_actnorm=True, ctx=ctx, context=context, cnoise=cn())
        for i in range(warmups + iters):
            noise = cn()
            t0 = time.perf_counter()
            fo.flow_forward(ops, params, x, u, eps, ctx=ctx, context=context, cnoise=noise)
            if i >= warmups:
                times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B / med, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "median of %d specialist forward passes (oracle/flow_oracle.py, contexts %s, %s + %s, contextflow) over one batch of %d "
                      "after %d warm-up, torch %d threads" % (iters, ctx["contexts"], ctx["enc_emb"], ctx["enc_type"], B, warmups, cores)}


def secondary_specialist(name, dev, B, iters, cpu):
    """the context-conditioned (specialist) forward under --contextflow (README: generalist frozen, one
    context encoder + CN net per layer => per-sample Conv1x1 / ActNorm / conditioner bias, context-shifted mixtures)."""
    import contextflow_amd as cfa
    ctx = SPECIALIST_CTX[name]
    torch.manual_seed(0)
    cfg, ds, M = cfa.preset_config(name)
    cfg.update(generalist=False, enc_emb=ctx["enc_emb"], enc_type=ctx["enc_type"], contextflow=ctx["contextflow"])
    model = cfa.create_model(cfg, ds, M, contexts=ctx["contexts"]).to(dev).eval()
    for p in model.parameters():                       # CN nets are zero-initialised (model.py): perturb so that they do something
        if p.abs().max() == 0:
            p.data.normal_(0, 0.02)
    x = synth(name, B, dev, seed=5000)
    context = torch.stack([torch.randint(0, k, (B,), device=dev) for k in ctx["contexts"]], 1)
    ev = []
    with torch.no_grad():
        for _ in range(2):
            model(x, context)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            _, logp = model(x, context)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    from contextflow_amd.layers import _hip
    L = _hip.lib()
    levels = [(16, 16, 4), (32, 8, 4), (64, 4, 4)]
    alg = total_flop_per_sample(name)
    exe = sum(2 * n * L.cf_flow_step_macs(B, C, H, H, 0) for C, H, n in levels)
    tf_alg, tf_exe = alg * B / dt / 1e12, exe * B / dt / 1e12
    out = {"metric": "samples/s specialist fwd+logdet (--contextflow), %s" % LABEL[name], "value": round(B / dt, 1), "unit": "samples/s",
           "config": {"workload": "%s --coupling %s --contextflow, enc %s + %s, contexts %s" % (name, cfg["coupling"], ctx["enc_emb"], ctx["enc_type"], ctx["contexts"]),
                      "batch": B, "iters": iters},
           "ms_per_step": round(dt * 1e3, 3), "finite": bool(torch.isfinite(logp).all()),
           "roofline": {"bound": "mfma", "achieved": round(tf_exe, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf_exe / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                        "algorithmic_tflops": round(tf_alg, 2), "algorithmic_speedup": round(tf_alg / tf_exe, 3),
                        "basis": "whole call: dense multiply-adds executed by the matrix pipe / wall time"}}
    if cpu:
        out["cpu_baseline"] = cpu_specialist_baseline(name)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    del model, x
    torch.cuda.empty_cache()
    return out


def secondary_specialist_training(name, dev, B, iters):
    """SURVEY.md 8(f)2, training: one eager training step of the specialist flow under --contextflow (README: the generalist is frozen,
    the CN nets train) - forward through the grouped context front end, hand-written backward, torch.optim.AdamW; the loss of
    experiment_cl.py:128-133's classification term.  No CPU leg (the oracle has no specialist backward)."""
    import contextflow_amd as cfa
    ctx = SPECIALIST_CTX[name]
    torch.manual_seed(0)
    cfg, ds, M = cfa.preset_config(name)
    cfg.update(generalist=False, enc_emb=ctx["enc_emb"], enc_type=ctx["enc_type"], contextflow=ctx["contextflow"])
    model = cfa.create_model(cfg, ds, M, contexts=ctx["contexts"]).to(dev)
    x = synth(name, B, dev, seed=5100)
    gt = torch.randint(0, M, (B,), device=dev)
    context = torch.stack([torch.randint(0, k, (B,), device=dev) for k in ctx["contexts"]], 1)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3)
    dim_inv = 1.0 / (ds[0] * ds[1] * ds[2])

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(dim_inv * model.log_prob(x, context), gt)
        loss.backward()
        opt.step()
        return loss.detach()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    out = {"metric": "samples/s specialist training step (--contextflow: fwd + bwd + AdamW, eager), %s" % LABEL[name], "value": round(B / dt, 1),
           "unit": "samples/s",
           "config": {"workload": "%s --coupling %s --contextflow, enc %s + %s, contexts %s" % (name, cfg["coupling"], ctx["enc_emb"], ctx["enc_type"], ctx["contexts"]),
                      "batch": B, "iters": iters, "trainable_tensors": len(params), "launch": "eager"},
           "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss), 5), "finite": bool(torch.isfinite(loss))}
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


def train_main(a, model, cfg, name, dev, rank, local_rank, world, rehearsal, single_pg, t_start, real_stdout):
    """--train: the data-parallel training step (experiment_cl.py:123-136 on N ranks).  Every rank owns --train-batch samples
    (weak scaling) and replays ONE captured HIP graph per step: forward, the reference's loss, hand-written backward whose
    gradient kernels write into the flat bucket that p.grad views, one RCCL all-reduce per bucket segment enqueued as soon as
    the segment is complete (it overlaps the backward of the levels below), fused AdamW behind the last of them."""
    import torch.distributed as dist
    B = a.train_batch
    x = synth(name, B, dev, seed=4000 + rank)
    M = model.mixtures
    gt = torch.randint(0, M, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank))
    loss_fn = reference_loss(name)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
    if rehearsal:                    # gloo moves CUDA tensors through the host: not capturable - the rehearsal runs the same step eagerly
        model.data_parallel = True

        def step(xb, yb):
            opt.zero_grad(set_to_none=True)
            l = loss_fn(model.log_prob(xb), yb)
            l.backward()
            opt.step()
            return l.detach()
    else:
        step = model.capture_train_step(x, loss_fn, opt)
    for _ in range(max(a.warmup, 2)):
        step(x, gt)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(x, gt)
    fence()
    dt = time.perf_counter() - t0
    table = torch.zeros(world, 3, dtype=torch.float64, device=dev)
    table[rank] = torch.tensor([dt, float(loss), float(local_rank)], dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        alg, exe = train_flop_per_sample(name, B)
        tf_exe, tf_alg = exe * B * world * a.steps / dt / 1e12 / world, alg * B * world * a.steps / dt / 1e12 / world
        bucket = model._grad_bucket
        out = {"metric": "samples/s training step (fwd + bwd + gradient all-reduce + AdamW), %s" % LABEL[name],
               "value": round(B * world * a.steps / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": max(a.warmup, 2),
               "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s --coupling %s, training step %s" % (name, cfg["coupling"], "launched eagerly (gloo rehearsal)" if rehearsal else "as one captured HIP graph per rank"),
                          "global_batch": B * world, "per_gpu_batch": B, "parallelism": "dp%d" % world,
                          "collective": "all_reduce(gradient bucket segment) x %d per step, fp32, inside the graph" % (len(bucket.segments) if bucket else 0),
                          "loss": "experiment_ad.py:207" if M == 1 else "experiment_cl.py:128-133"},
               "roofline": {"bound": "mfma", "achieved": round(tf_exe, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf_exe / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None, "algorithmic_tflops": round(tf_alg, 2),
                            "basis": "per GPU, whole step: dense multiply-adds (forward, data gradients, weight gradients) / wall time"},
               "comm": {"backend": dist.get_backend() if dist.is_initialized() else None, "world_size": world,
                        "data_parallel": bool(model.data_parallel), "single_rank_communicator": single_pg, "rehearsal_gloo_on_one_device": rehearsal,
                        "gradient_messages_bytes": bucket.message_bytes() if bucket else [],
                        "per_rank": [{"rank": r, "local_rank": int(v[2]), "wall_s": round(float(v[0]), 4), "last_loss": round(float(v[1]), 5)}
                                     for r, v in enumerate(table.cpu())]},
               "bench_wall_s": round(time.perf_counter() - t_start, 1)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


def main():
    a = parse()
    import contextflow_amd as cfa
    from contextflow_amd import dist as cdist
    from contextflow_amd.layers import _hip
    import torch.distributed as dist

    t_start = time.perf_counter()
    # ONE JSON line on stdout: everything else any library writes there (RCCL prints a version banner when its first
    # communicator comes up) goes to stderr - file descriptor 1 points at stderr until the line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
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
    # N = 1: the headline line goes through a real RCCL communicator of ONE rank (CF_DIST_SINGLE_RANK=1: dist._active), so the
    # driver's single-GPU run exercises the collectives of the data-parallel path - flat parameter broadcast, fp64 NLL
    # all-reduce - exactly as an N-rank run does.  BENCH_NO_SINGLE_RANK_PG=1 skips it (A/B of its cost: none measurable).
    single_pg = world == 1 and not rehearsal and os.environ.get("BENCH_NO_SINGLE_RANK_PG") != "1"
    if single_pg:
        os.environ["CF_DIST_SINGLE_RANK"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    else:
        cdist.init_process_group("gloo" if rehearsal else "nccl")
    _hip.lib()
    name = a.workload

    # model: reference init under a fixed seed; rank 0 runs the ActNorm data-dependent init, then ONE flat broadcast
    model, cfg = build(name, dev, rank)
    cdist.broadcast_parameters(model, src=0)

    if a.train:
        train_main(a, model, cfg, name, dev, rank, local_rank, world, rehearsal, single_pg, t_start, real_stdout)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    G = a.global_batch
    lo, hi = cdist.shard_bounds(G, rank, world)
    x = synth(name, hi - lo, dev, seed=1000 + rank)
    nll_acc = torch.zeros(1, dtype=torch.float64, device=dev)
    ar_events = []                                      # (start, end) HIP events around the all-reduce of every timed step

    @torch.no_grad()                                   # density evaluation (experiment_cl.py:163-185 runs it under no_grad):
    def step(timed):                                   # no autograd tape, no W^-1 for the backward
        nll_acc.zero_()
        for c0 in range(0, hi - lo, a.chunk):
            xb = x[c0:c0 + a.chunk]
            _, logp = model(xb)
            _hip.call("cf_nll_sum", _hip.p(logp), _hip.p(nll_acc), logp.shape[0], logp.shape[1], _hip.stream())
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        red = cdist.allreduce_nll(nll_acc, hi - lo)            # RCCL all-reduce of [sum log p, count]
        if timed:
            e1.record()
            ar_events.append((e0, e1))
        return red

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
    torch.cuda.synchronize()
    from contextflow_amd.layers import coupling as _cpl
    model.step_events, _cpl.VIT_EVENTS = None, None
    # per-rank breakdown (the first multi-GPU run should be diagnosable from its one JSON line): wall time of the timed
    # region, time inside the dominant kernel, time between "my last kernel is done" and "the all-reduce is done" (RCCL
    # latency + waiting for the slowest rank), gathered on rank 0
    kern_s = sum(e[0].elapsed_time(e[1]) for e in events) * 1e-3
    ar_ms = [e[0].elapsed_time(e[1]) for e in ar_events]
    mine = torch.tensor([dt, kern_s, sum(ar_ms) * 1e-3, max(ar_ms) * 1e-3 if ar_ms else 0.0, float(hi - lo), float(local_rank)],
                        dtype=torch.float64, device=dev)
    table = torch.zeros(world, mine.numel(), dtype=torch.float64, device=dev)
    table[rank] = mine
    if world > 1:
        dist.all_reduce(table, op=dist.ReduceOp.SUM)          # a gather written as a sum of one-hot rows (works on RCCL and gloo)
    allr = list(table)
    per_rank = [{"rank": r, "local_rank": int(v[5]), "samples_per_step": int(v[4]), "wall_s": round(float(v[0]), 4),
                 "kernel_s": round(float(v[1]), 4), "allreduce_s": round(float(v[2]), 5), "allreduce_max_ms": round(float(v[3]) * 1e3, 3)}
                for r, v in enumerate(t.cpu() for t in allr)]
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    bpd = cdist.mean_bits_per_dim(red.cpu(), DIMS[name])
    roof = roofline(events, name, dt)
    walls, kerns, ars = [p["wall_s"] for p in per_rank], [p["kernel_s"] for p in per_rank], [p["allreduce_s"] for p in per_rank]
    comm = {"backend": dist.get_backend() if dist.is_initialized() else None, "world_size": world,
            "world_size_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "visible_devices": torch.cuda.device_count(),
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
            "rehearsal_gloo_on_one_device": rehearsal, "single_rank_communicator": single_pg,
            "wall_s_min_max": [min(walls), max(walls)], "kernel_s_min_max": [min(kerns), max(kerns)],
            "allreduce_s_min_max": [min(ars), max(ars)], "allreduce_share_of_step": round(max(ars) / dt, 5),
            "per_rank": per_rank}

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
                secondary_training("cifar10", dev, 16384, 10, graph=False),            # (before the leg that ends with the CPU baseline:
                secondary_training("cifar10", dev, 16384, 10, graph=True, cpu=cpu),    #  its idle worker threads slow an eager launcher down)
                secondary_training("cifar10", dev, 256, 50, graph=True),
                secondary_training("cifar10", dev, 256, 50, graph=True, own_adamw=True),
                secondary_training("smap", dev, 32768, 10, graph=True, cpu=cpu),       # BASELINE config 4's training step
                secondary_training("smap", dev, 256, 50, graph=True),                  # ... at the reference's batch (config.py:10)
                secondary_training("smap", dev, 256, 50, graph=True, own_adamw=True),
                secondary_specialist("cifar10", dev, 32768, 5, cpu),                   # --contextflow specialist forward
                secondary_specialist_training("cifar10", dev, 8192, 5),                # ... and its training step
                secondary_sampling("mnist", dev, 16384, 10, cpu),                      # SURVEY 8(f)3 / north_star "forward+inverse": flow.sample
                secondary_sampling("cifar10", dev, 16384, 10, cpu=False),              # (SplitPrior.reverse by specification: no reference / oracle chain)
            ]
        out["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
