#!/usr/bin/env python3
"""Evaluation latency at the reference's batch sizes with and without chained flow steps (cf_flow_step_fwd_chain): eager and
auto-graph-replayed `flow.log_prob(x)`.  usage: chain_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from contextflow_amd.layers.flowsequential import FlowSequential
dev = torch.device("cuda", 0)
for name, B in (("cifar10", 64), ("cifar10", 256), ("cifar10", 512), ("mnist", 64), ("mnist", 256), ("smap", 64), ("smap", 256), ("smap", 1024)):
    out = []
    for chain in (False, True):
        FlowSequential.CHAIN_STEPS = chain
        model, cfg = bench.build(name, dev)
        x = bench.synth(name, B, dev, seed=3000)
        res = {}
        with torch.no_grad():
            for mode in ("eager", "graph"):
                model.auto_graph = mode == "graph"
                if mode == "graph":
                    model._replay_wins = lambda g, inp: True
                for _ in range(20):
                    model.log_prob(x)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(300):
                    model.log_prob(x)
                torch.cuda.synchronize()
                res[mode] = (time.perf_counter() - t0) / 300 * 1e6
        out.append("%s: eager %.1f us, graph %.1f us" % ("chained" if chain else "one launch per step", res["eager"], res["graph"]))
    print("%s B=%d | %s" % (name, B, " | ".join(out)), flush=True)
FlowSequential.CHAIN_STEPS = True
