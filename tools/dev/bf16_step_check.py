#!/usr/bin/env python3
"""The bf16-piece forms of the 16x16 level's step kernel (G16wb: Winograd-domain products, debug variant 6 / CONTEXTFLOW_BF16_SPLIT=1;
G16db: direct 3x3 with h1 split by its producer, variant 7 / CONTEXTFLOW_BF16_SPLIT=2) against the fp32 Winograd form (variant 4): z and the log-det of the same step on the same input, and both kernel times.
usage: bf16_step_check.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import _hip
L = cfa.layers
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lib = _hip.lib()
fn = lib.cf_flow_step_fwd_debug
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
C, H, W = 16, 16, 16
torch.manual_seed(0)
conv, act, cpl = L.Conv1x1((C, H, W)).to(dev), L.ActNorm((C, H, W)).to(dev), L.Coupling(C, (3, 3), (1, 1)).to(dev)
with torch.no_grad():
    for p in cpl.parameters():
        p.normal_(0, 0.08)
    act.NN_t.normal_(0, 0.1); act.NN_logs.normal_(0, 0.1)
x = torch.randn(B, C, H, W, device=dev)
ws = torch.empty(lib.cf_flow_step_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
lib.cf_bf16_split(1)
f, pp = _hip.f32, _hip.p
c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
_hip.call("cf_flow_step_prepare", pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
          pp(f(c1.weight.detach())), pp(f(c1.bias.detach())), pp(f(c2.weight.detach())), pp(f(c2.bias.detach())),
          pp(f(c3.weight.detach())), pp(f(c3.bias.detach())), pp(ws), C, H, W, _hip.stream())
res = {}
for name, var in (("fp32 MFMA (variant 4)", 4), ("bf16 pieces (variant 6)", 6), ("direct bf16 (variant 7)", 7), ("direct (variant 3)", 3)):
    z = torch.full_like(x, float("nan"))
    ldj = torch.zeros(B, device=dev)
    flags = var << 16
    for _ in range(3):
        ldj.zero_()
        _hip.check(fn(pp(x), pp(z), pp(ldj), pp(ws), B, C, H, W, C * H * W, 0, None, flags, _hip.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _hip.check(fn(pp(x), pp(z), pp(ldj), pp(ws), B, C, H, W, C * H * W, 0, None, flags, _hip.stream()))
    e1.record(); torch.cuda.synchronize()
    ldj.zero_()
    _hip.check(fn(pp(x), pp(z), pp(ldj), pp(ws), B, C, H, W, C * H * W, 0, None, flags, _hip.stream()))
    torch.cuda.synchronize()
    res[name] = (z.clone(), ldj.clone(), e0.elapsed_time(e1) / 20)
    print("%-26s %.3f ms per %d samples, finite: %s" % (name, res[name][2], B, bool(torch.isfinite(z).all())))
zr, lr, _ = res["direct (variant 3)"]
for name in ("fp32 MFMA (variant 4)", "bf16 pieces (variant 6)", "direct bf16 (variant 7)"):
    z, l, _ = res[name]
    print("%-26s vs the direct form: max |dz| %.3e (|z| max %.2f), max |d ldj| %.3e, rms d ldj %.3e, bits/dim of the worst sample %.3e"
          % (name, (z - zr).abs().max().item(), zr.abs().max().item(), (l - lr).abs().max().item(), (l - lr).pow(2).mean().sqrt().item(),
             (l - lr).abs().max().item() / (3072 * 0.6931)))
