#!/usr/bin/env python3
"""Kernel time of the wave form of the transformer flow step (cf_vit_step_fwd) at a saturating batch, 20 back-to-back launches
between HIP events; CONTEXTFLOW_HIP_LIB selects an ablation build (tools/dev/make_abl.py).  usage: vit_wave_time.py [B=524288]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config("smap")
model = cfa.create_model(cfg, ds, M).to(dev)
torch.set_grad_enabled(False)
model(torch.rand(256, *ds, device=dev))
conv, act, cpl = model.sequence_modules[1:4]
x = torch.randn(B, 26, 8, 1, device=dev)
ld = torch.zeros(B, device=dev)
ws = cpl.step_prepare(conv.NN, act.NN_t, act.NN_logs, dev, "wave")
for _ in range(3):
    cpl.step_forward(x, ws, ld, variant="wave")
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        cpl.step_forward(x, ws, ld, variant="wave")
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / n)
flop = 944768.0 * B
print("%-28s wave step B=%d: %.1f us  %.1f TFLOP/s algorithmic = %.3f of 157.3" % (os.path.basename(os.environ.get("CONTEXTFLOW_HIP_LIB", "product")), B, best, flop / best * 1e-6, flop / best * 1e-6 / 157.3), flush=True)
