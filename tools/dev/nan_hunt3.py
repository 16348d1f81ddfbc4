import os, sys, weakref
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import _hip, autograd as ag
B = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config("cifar10")
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
with torch.no_grad():
    model(x[:256])
live = {}
orig_empty = torch.empty
def empty(*a, **k):
    t = orig_empty(*a, **k)
    if t.is_cuda:
        live[t.data_ptr()] = weakref.ref(t)
        if t.dim() == 3 and t.dtype == torch.float32:
            t.fill_(7777.0)
    return t
torch.empty = empty
orig_call = _hip.call
found = [0]
def call(name, *a):
    r = orig_call(name, *a)
    if name == "cf_wgrad" and found[0] < 3:
        torch.cuda.synchronize()
        ptr = lambda v: v.value if hasattr(v, "value") else int(v)
        A, Bm, gw = (live.get(ptr(a[i])) for i in (0, 1, 2))
        A, Bm, gw = (t() if t is not None else None for t in (A, Bm, gw))
        if A is not None and (A == 7777.0).any():
            idx = (A == 7777.0).nonzero()
            print("UNWRITTEN in A: count %d first %s last %s shape %s taps %s" % (idx.shape[0], idx[0].tolist(), idx[-1].tolist(), tuple(A.shape), a[10]))
            found[0] += 1
        if gw is not None and (not torch.isfinite(gw).all() or gw.abs().max() > 1e4):
            found[0] += 1
            msg = "BAD cf_wgrad B=%s MR=%s NR=%s H=%s taps=%s |gw|max %.3g" % (a[5], a[6], a[7], a[8], a[10], gw[torch.isfinite(gw)].abs().max().item())
            for nm, t in (("A", A), ("Bm", Bm)):
                if t is None: msg += " %s: untracked" % nm
                else:
                    fin = torch.isfinite(t)
                    big = (t.abs() > 1e4) & fin
                    msg += " | %s shape %s nonfinite %d huge %d" % (nm, tuple(t.shape), int((~fin).sum()), int(big.sum()))
                    if (~fin).any() or big.any():
                        idx = ((~fin) | big).nonzero()
                        msg += " first %s last %s" % (idx[0].tolist(), idx[-1].tolist())
                        b0 = idx[0][0].item()
                        bs = sorted(set(idx[:, 0].tolist()))
                        msg += "\n   bad samples %s" % bs[:24]
                        rows = sorted(set(idx[idx[:, 0] == b0][:, 1].tolist())); pixs = sorted(set(idx[idx[:, 0] == b0][:, 2].tolist()))
                        msg += "\n   sample %d: bad rows %s pixels %s" % (b0, rows, pixs)
                        msg += "\n   values row %d: %s" % (rows[0], t[b0, rows[0]].tolist())
                        msg += "\n   values row %d (neighbour, fine): %s" % (rows[0] - 1, t[b0, rows[0] - 1].tolist()[:8])
            print(msg)
    return r
_hip.call = call
for it in range(6):
    model.zero_grad(set_to_none=True)
    _, lp = model(x)
    torch.nn.functional.cross_entropy(lp / 3072.0, gt).backward()
    torch.cuda.synchronize()
print("done, found", found[0])
