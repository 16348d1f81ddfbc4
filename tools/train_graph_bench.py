#!/usr/bin/env python3
"""Whole training step (forward + hand-written backward + AdamW) captured into ONE HIP graph with torch.cuda.graphs
(launch-bound regimes: the SMAP transformer flow runs ~1800 small kernels per step).  usage: [name] [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa

name = sys.argv[1] if len(sys.argv) > 1 else "smap"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.rand(B, *ds, device=dev) if M == 1 else torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
with torch.no_grad():
    model(x[:256])
opt = (cfa.optim.FusedAdamW(model.parameters(), lr=1e-4) if os.environ.get("CF_OWN_ADAMW") == "1"      # CF_OWN_ADAMW=1: contextflow_amd.optim
       else torch.optim.AdamW(model.parameters(), lr=1e-4, capturable=True, fused=True))      # one multi-tensor kernel per step
dim_inv = 1.0 / (ds[0] * ds[1] * ds[2])


def step():
    opt.zero_grad(set_to_none=True)       # the engine takes the buffers the backward returns: no fill / accumulate launches
    logp = dim_inv * model.log_prob(x)
    loss = -logp.mean() if M == 1 else torch.nn.functional.cross_entropy(logp, gt)
    loss.backward()
    opt.step()
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / iters
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
l0 = float(loss)
t0 = time.perf_counter()
for _ in range(iters):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / iters
print("%s B=%d: eager %.2f ms = %.0f samples/s; graph replay %.2f ms = %.0f samples/s; loss %.4f -> %.4f" % (
    name, B, eager * 1e3, B / eager, graph * 1e3, B / graph, l0, float(loss)))
