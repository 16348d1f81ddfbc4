#!/usr/bin/env python3
"""Per-kernel timings of the layer-by-layer ViT kernels at the ATM level-1 shape (4096 samples x 36 tokens, width 152).
usage: vit_kernels_bench.py [B] [ntok] [dim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextflow_amd.layers import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 36
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 152
dev = "cuda:0"
rows, dh = B * ntok, 64
L, st, p = _hip.lib(), _hip.stream, _hip.p


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


x = torch.randn(rows, dim, device=dev)
gy = torch.randn(rows, dim, device=dev)
W = torch.randn(dim, dim, device=dev) / dim ** 0.5
bias = torch.randn(dim, device=dev)
y = torch.empty(rows, dim, device=dev)
t = timeit(lambda: _hip.call("cf_linear", p(x), p(W), p(bias), None, p(y), rows, dim, dim, 0, st()))
fl = 2.0 * rows * dim * dim
print("cf_linear        %dx%d @ %dx%d: %7.1f us  %5.1f TFLOP/s" % (rows, dim, dim, dim, t, fl / t / 1e6))
gW, gb = torch.empty(dim, dim, device=dev), torch.empty(dim, device=dev)
ws = torch.empty(L.cf_linear_wgrad_ws_bytes(rows, dim, dim), device=dev, dtype=torch.uint8)
t = timeit(lambda: _hip.call("cf_linear_wgrad", p(x), p(gy), p(gW), p(gb), p(ws), rows, dim, dim, st()))
print("cf_linear_wgrad  (+reduce)         : %7.1f us  %5.1f TFLOP/s" % (t, fl / t / 1e6))
t = timeit(lambda: gy.t() @ x)
print("library gy^T x                     : %7.1f us  %5.1f TFLOP/s" % (t, fl / t / 1e6))
lw = torch.randn(dim, device=dev)
gx = torch.empty_like(x)
part = torch.empty(L.cf_layernorm_bwd_parts(), 2 * dim, device=dev)
t = timeit(lambda: _hip.call("cf_layernorm_bwd", p(x), p(lw), p(gy), p(gx), p(part), rows, dim, 1e-5, st()))
print("cf_layernorm_bwd                   : %7.1f us  %5.2f TB/s (x, gy, gx)" % (t, 3.0 * rows * dim * 4 / t / 1e6))
t = timeit(lambda: _hip.call("cf_layernorm", p(x), p(lw), p(bias), None, p(y), rows, dim, ntok, 1e-5, st()))
print("cf_layernorm                       : %7.1f us  %5.2f TB/s (x, y)" % (t, 2.0 * rows * dim * 4 / t / 1e6))
qkv = torch.randn(rows, 3 * dh, device=dev)
o = torch.empty(rows, dh, device=dev)
t = timeit(lambda: _hip.call("cf_attention", p(qkv), p(o), B, ntok, dh, dh ** -0.5, st()))
print("cf_attention     N=%d dh=%d        : %7.1f us  %5.2f TFLOP/s" % (ntok, dh, t, 4.0 * B * ntok * ntok * dh / t / 1e6))
go, gq = torch.randn(rows, dh, device=dev), torch.empty(rows, 3 * dh, device=dev)
t = timeit(lambda: _hip.call("cf_attention_bwd", p(qkv), p(go), p(gq), B, ntok, dh, dh ** -0.5, st()))
print("cf_attention_bwd                   : %7.1f us  %5.2f TFLOP/s" % (t, 12.0 * B * ntok * ntok * dh / t / 1e6))
