#!/usr/bin/env python3
"""Dev check (GPU): bench.py's EAGER training leg on its own (no captured leg before it), twice.  usage: eager_train_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for graph in (False, True, False):
    r = bench.secondary_training("cifar10", torch.device("cuda:0"), B, 10, graph)
    print("graph" if graph else "eager", r["ms_per_step"], r["value"], flush=True)
