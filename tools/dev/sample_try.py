import sys; sys.path.insert(0, "/root/repo")
import torch, traceback
import contextflow_amd as cfa
dev="cuda:0"
def try_(name, spec):
    torch.manual_seed(0)
    cfg, ds, M = cfa.preset_config(name)
    ctxs = {"cifar10":[15,5],"mnist":[64],"smap":[55],"atm":[68]}[name]
    kw = {}
    if spec:
        cfg.update(generalist=False, enc_emb="onehot", enc_type="uniform", contextflow=True); kw=dict(contexts=ctxs)
    m = cfa.create_model(cfg, ds, M, **kw).to(dev).eval()
    x = torch.rand(64, *ds, device=dev) if name in ("smap","atm") else torch.randint(0,256,(64,*ds),device=dev).float()
    ctx = torch.stack([torch.randint(0,k,(64,),device=dev) for k in ctxs],1) if spec else None
    with torch.no_grad():
        m(x, ctx)
        try:
            out = m.sample(8, ctx[:8] if spec else None)
            out = out[0] if isinstance(out, tuple) else out
            print(name, "specialist" if spec else "generalist", "sample ok", tuple(out.shape), bool(torch.isfinite(out).all()))
        except Exception as e:
            print(name, "specialist" if spec else "generalist", "sample FAILED:", type(e).__name__, str(e)[:150])
for n in ("mnist","cifar10","smap","atm"):
    for s in (False, True):
        try_(n, s)
