import os, sys
sys.path.insert(0, "/root/repo")
import torch
from contextflow_amd.layers import _hip
dev="cuda:0"; L,st,p=_hip.lib(),_hip.stream,_hip.p
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for rows in (32768, 65536, 131072, 147456, 262144, 524288):
    for K,N in ((152,152),(152,192),(64,152),(128,96)):
        x=torch.randn(rows,K,device=dev); W=torch.randn(N,K,device=dev); b=torch.randn(N,device=dev); y=torch.empty(rows,N,device=dev)
        t=timeit(lambda: _hip.call("cf_linear",p(x),p(W),p(b),None,p(y),rows,K,N,0,st()))
        print("rows %7d K %3d N %3d: %7.1f us %5.1f TF  %5.2f TB/s" % (rows,K,N,t,2.0*rows*K*N/t/1e6,(rows*(K+N)*4)/t/1e6))
