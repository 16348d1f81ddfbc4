import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, time, numpy as np
from contextflow_amd.layers import _hip
dev='cuda'
for C in (4,8,12,16,32,64,76,128):
    g=torch.Generator().manual_seed(C)
    W=torch.linalg.qr(torch.randn(C,C,generator=g))[0]+0.05*torch.randn(C,C,generator=g)
    W=W.contiguous(); Wd=W.to(dev); lad=torch.empty(1,device=dev); inv=torch.empty(C,C,device=dev)
    for want in ((False,) if C > 64 else (False,True)):
        for _ in range(3): _hip.call("cf_slogdet_inverse",_hip.p(Wd),C,_hip.p(lad),_hip.p(inv) if want else None,_hip.stream())
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): _hip.call("cf_slogdet_inverse",_hip.p(Wd),C,_hip.p(lad),_hip.p(inv) if want else None,_hip.stream())
        e1.record(); torch.cuda.synchronize()
        ref=torch.linalg.slogdet(W.double())[1].item()
        err=abs(lad.item()-ref)
        ierr=(inv.cpu().double()-torch.linalg.inv(W.double())).abs().max().item() if want else 0
        print(f"C={C} inv={want} {e0.elapsed_time(e1)/50*1e3:.1f} us  lad_err={err:.2e} inv_err={ierr:.2e}")
