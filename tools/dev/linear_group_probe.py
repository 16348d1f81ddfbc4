"""cf_linear_group on the CN-net shapes of the cifar10 specialist flow against a plain fill of its outputs (developer probe)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
from contextflow_amd.layers import _hip
B, K = 32768, 20
dev = "cuda:0"
ctx = torch.stack([torch.randint(0, 15, (B,)), torch.randint(0, 5, (B,))], 1).to(dev)
card = torch.tensor([15, 5], device=dev)
def run(Ns, label):
    n = len(Ns)
    us = [torch.rand(B, K, device=dev) for _ in Ns]; qs = [torch.ones(K, device=dev) for _ in Ns]
    Ws = [torch.randn(N, K, device=dev) for N in Ns]; bs = [torch.randn(N, device=dev) for N in Ns]
    ys = [torch.empty(B, N, device=dev) for N in Ns]
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts]); iarr = lambda v: (ctypes.c_int * n)(*v)
    f = lambda: _hip.call("cf_linear_group", n, arr(us), arr(qs), arr(Ws), arr(bs), arr(ys), None, iarr(Ns), iarr([0] * n), _hip.p(ctx), _hip.p(card), 2, 1, B, K, _hip.stream())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    byt = sum(B * N * 4 for N in Ns)
    g = lambda: _hip.call("cf_linear_group", n, arr(us), None, arr(Ws), arr(bs), arr(ys), None, iarr(Ns), iarr([0] * n), None, None, 0, 0, B, K, _hip.stream())
    g(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): g()
    e1.record(); torch.cuda.synchronize()
    print("   plain inputs (no code formed): %.3f ms" % (e0.elapsed_time(e1) / 10))
    e0.record()
    for _ in range(10):
        for y in ys: y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / 10
    print("%s: group %.3f ms (%.2f TB/s written), fill of the same outputs %.3f ms (%.2f TB/s)" % (label, t, byt / t / 1e9, tf, byt / tf / 1e9))
run([2560] * 4, "4 x N=2560")
run([768] * 4, "4 x N=768")
run([256] * 4, "4 x N=256")
run([128] * 8 + [64] * 8 + [32] * 8, "small")
