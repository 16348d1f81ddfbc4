import sys; sys.path.insert(0,'/root/repo')
import torch
from contextflow_amd.layers.simple_vit import _linear
torch.manual_seed(0)
for K,N,rows in ((256,128,37),(128,4096,5),(128,64,300),(64,32,4),(200,40,9)):
    lin=torch.nn.Linear(K,N).cuda()
    x=torch.randn(rows,K,device='cuda')
    y=_linear(x,lin)
    ref=x.double()@lin.weight.double().t()+lin.bias.double()
    print(K,N,rows,(y.double()-ref).abs().max().item())
    if K<=128:
        y2=_linear(x,lin,act=2)
        print("  relu",(y2.double()-ref.clamp(min=0)).abs().max().item())
