#!/usr/bin/env python3
"""Small-batch latency: eager fused forward vs HIP-graph replay.  usage: latency_bench.py [name] [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa

name = sys.argv[1] if len(sys.argv) > 1 else "mnist"
Bs = [int(v) for v in sys.argv[2:]] or [64, 256, 1024]
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev)
torch.set_grad_enabled(False)                       # evaluation (experiment_cl.py:163-185)
model.auto_graph = False                            # "eager" = every kernel launched by the host; "graph" = explicit capture
model(torch.randint(0, 256, (256, *ds), device=dev).float())
for B in Bs:
    x = torch.randint(0, 256, (B, *ds), device=dev).float()
    g = model.capture(x)
    for fn, tag in ((lambda: model(x), "eager"), (lambda: g(x), "graph")):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 50
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%s B=%d %s: %.1f us/call, %.0f samples/s" % (name, B, tag, dt * 1e6, B / dt), flush=True)
