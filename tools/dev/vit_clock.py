#!/usr/bin/env python3
"""Shader clock and per-wave lifetime of the wave form of the transformer flow step (probe build:
tools/dev/make_abl.py ticks --only=cf_vit_step.hip -DCF_VS_TICKS [-DCF_ABL_VS_...]): every wave reports the shader cycles
(s_memtime) and 100 MHz reference ticks (s_memrealtime) between its first and last instruction.  usage: vit_clock.py [B=524288]"""
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
h = torch.zeros(B, 26, 8, 1, device=dev)
for _ in range(5):
    cpl.step_forward(x, ws, ld, variant="wave")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
cpl.step_forward(x, ws, ld, h_out=h, variant="wave")
e1.record()
torch.cuda.synchronize()
nw = B // 8
t = h.flatten()[: 2 * nw].view(nw, 2).double().cpu()
cyc, ref = t[:, 0], t[:, 1]
ghz = (cyc / (ref / 100e6)).median().item() * 1e-9
print("%-24s kernel %.0f us; per wave: %.0f shader cycles, %.1f us alive (median), shader clock %.2f GHz; 1326 MFMAs x 64 cycles = %.0f%% of a wave's life / its share of a SIMD with 2 waves resident: %.0f%%" % (
    os.path.basename(os.environ.get("CONTEXTFLOW_HIP_LIB", "product")), e0.elapsed_time(e1) * 1e3, cyc.median(), (ref / 100).median(), ghz,
    100 * 1326 * 64 / cyc.median(), 200 * 1326 * 64 / cyc.median()))
