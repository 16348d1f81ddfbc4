#!/usr/bin/env python3
"""Kernel time of ONE transformer flow step (SMAP geometry) per batch size in its two one-kernel forms (rs: row-split
workgroups of 4 samples, cf_vit_step_rs_fwd; wave: 8 samples per wave, cf_vit_step_fwd): where does the cross-over sit?
20 back-to-back launches between HIP events.  usage: vit_variants.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa

Bs = [int(v) for v in sys.argv[1:]] or [64, 256, 1024, 2048, 4096, 8192, 16384, 65536]
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config("smap")
model = cfa.create_model(cfg, ds, M).to(dev)
torch.set_grad_enabled(False)
model(torch.rand(256, *ds, device=dev))
conv, act, cpl = model.sequence_modules[1:4]
C = 26
for B in Bs:
    x = torch.randn(B, C, 8, 1, device=dev)
    ld = torch.zeros(B, device=dev)
    res = []
    for variant in ("rs", "wave"):
        ws = cpl.step_prepare(conv.NN, act.NN_t, act.NN_logs, dev, variant)
        for _ in range(3):
            cpl.step_forward(x, ws, ld, variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            cpl.step_forward(x, ws, ld, variant=variant)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        res.append("%s %.1f us (%.1f M samples/s)" % (variant, us, B / us))
    print("step B=%d: %s" % (B, " | ".join(res)), flush=True)
