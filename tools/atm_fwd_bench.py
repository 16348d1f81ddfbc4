#!/usr/bin/env python3
"""Forward throughput of a generalist time-series preset (transformer couplings).  usage: atm_fwd_bench.py [name] [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa

name = sys.argv[1] if len(sys.argv) > 1 else "atm"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev).eval()
x = torch.rand(B, *ds, device=dev)
with torch.no_grad():
    for _ in range(2):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        _, logp = model(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print("%s B=%d: forward %.2f ms = %.0f samples/s (finite %s)" % (name, B, dt * 1e3, B / dt, bool(torch.isfinite(logp).all())))
