#!/usr/bin/env python3
"""Taped vs recompute backward (tests/test_gpu_parity.py::test_taped_backward_equals_recompute's comparison) across the batch
sizes where the taped forward changes kernels (row-split up to 512 / 1024 samples at 8x8 / 4x4): worst relative gradient
difference per batch size.  The two forwards differ by fp32 rounding, so ReLU units within rounding of zero flip and their
(finite) gradient paths differ: the difference does not shrink with the batch.  usage: taped_thresholds.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.gpu_util import build_model, set_noise
from tests.helpers import load_e2e
from contextflow_amd.layers import flowsequential as fs
import oracle.flow_oracle as fo
DEV = "cuda:0"
name = "cifar10"
ops, _, M, params, fx = load_e2e(name)
C, H, W = fo.CONFIGS[name][0]
for B in [int(a) for a in sys.argv[1:]] or [70, 511, 513, 1025, 2049]:
    g = torch.Generator().manual_seed(33)
    x = torch.randint(0, 256, (B, C, H, W), generator=g).float()
    u = torch.rand(B, C, H, W, generator=g)
    eps = [torch.randn(B, 1, H, W, generator=g)]
    wts = torch.randn(B, M, generator=g).to(DEV)
    out = {}
    try:
        for taped in (True, False):
            fs.TAPE_PLANES = taped
            model = build_model(name, params)
            set_noise(model, u, eps)
            model.train()
            _, logp = model(x.to(DEV))
            (logp * wts).sum().backward()
            out[taped] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    finally:
        fs.TAPE_PLANES = True
    worst = sorted(((float((out[True][k] - out[False][k]).abs().max() / max(float(out[False][k].abs().max()), 1e-6)), k) for k in out[True]), reverse=True)[:4]
    print("B=%d:" % B, ", ".join("%s %.1e" % (k, e) for e, k in worst), flush=True)
