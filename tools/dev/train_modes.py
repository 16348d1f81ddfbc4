#!/usr/bin/env python3
"""Training step of the cifar10 flow, eager vs one captured HIP graph, at several batches.  usage: train_modes.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [8192, 16384]:
    for graph in (False, True):
        r = bench.secondary_training("cifar10", dev, B, 10, graph)
        print(B, "graph" if graph else "eager", r["value"], "samples/s", r["ms_per_step"], "ms", flush=True)
