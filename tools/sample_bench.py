#!/usr/bin/env python3
"""Sampling throughput (`flow.sample`: prior draw + fused inverse steps).  usage: sample_bench.py [name] [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa
name = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.randint(0, 256, (256, *ds), device=dev).float()
with torch.no_grad():
    model(x)
    for _ in range(3):
        s = model.sample(B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        s = model.sample(B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
print("%s sample B=%d: %.2f ms = %.0f samples/s (finite %s)" % (name, B, dt * 1e3, B / dt, torch.isfinite(s).all().item()))
