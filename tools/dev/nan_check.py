import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config("cifar10")
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
with torch.no_grad():
    model(x[:256])
    torch.manual_seed(1); _, lp_eval = model(x)
torch.manual_seed(1)
_, logp = model(x)
print("B", B, "train-forward vs eval logp: max diff", (logp.detach() - lp_eval).abs().max().item(), "finite", torch.isfinite(logp).all().item())
loss = torch.nn.functional.cross_entropy(logp / 3072.0, gt)
loss.backward()
bad = [(k, p.grad.shape) for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print("loss", float(loss), "non-finite grads:", len(bad), bad[:8])
from contextflow_amd.layers import flowsequential as fs
def grads(tape):
    fs.TAPE_PLANES = tape
    model.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    _, lp = model(x)
    torch.nn.functional.cross_entropy(lp / 3072.0, gt).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
ref = grads(False)
for it in range(4):
    g = grads(True)
    worst = max(((g[k] - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 1e-12), k) for k in ref)
    nf = [k for k in g if not torch.isfinite(g[k]).all()]
    print("taped run", it, "worst rel diff vs recompute", worst, "non-finite", nf[:4])
g2 = grads(False)
print("recompute repeat: max diff", max((g2[k] - ref[k]).abs().max().item() for k in ref), "non-finite", [k for k in g2 if not torch.isfinite(g2[k]).all()][:4])
