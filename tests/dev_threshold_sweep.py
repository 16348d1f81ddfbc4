#!/usr/bin/env python3
"""Dev check (GPU): the same seeded cifar10 / mnist forward at batch sizes around the dispatch thresholds of the step kernels,
once with the default dispatch and once with CONTEXTFLOW_DIRECT_CONV=1 (a second process: the switch is read once), compared
sample by sample in bits/dim.  usage: dev_threshold_sweep.py  (spawns itself twice)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIZES = [1, 63, 255, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4097, 8191]
if len(sys.argv) > 1:
    import torch
    import contextflow_amd as cfa
    out = {}
    for name in ("cifar10", "mnist"):
        torch.manual_seed(0)
        cfg, ds, M = cfa.preset_config(name)
        model = cfa.create_model(cfg, ds, M).to("cuda:0")
        model.auto_graph = False
        g = torch.Generator().manual_seed(1)
        xall = torch.randint(0, 256, (max(SIZES), *ds), generator=g).float().to("cuda:0")
        with torch.no_grad():
            model(xall[:256])
            for B in SIZES:
                torch.manual_seed(5)
                out["%s_%d" % (name, B)] = model(xall[:B])[1].cpu()
    torch.save(out, sys.argv[1])
    sys.exit(0)
import torch
import math
res = []
for tag, env in (("default", {}), ("direct", {"CONTEXTFLOW_DIRECT_CONV": "1"})):
    path = "/tmp/sweep_%s.pt" % tag
    subprocess.run([sys.executable, os.path.abspath(__file__), path], env=dict(os.environ, **env), check=True)
    res.append(torch.load(path))
worst = 0.0
for k in res[0]:
    dims = 3072 if k.startswith("cifar10") else 1024
    bpd = lambda lp: -torch.logsumexp(lp.double(), -1) / (dims * math.log(2.0))
    d = (bpd(res[0][k]) - bpd(res[1][k])).abs().max().item()
    worst = max(worst, d)
    print("%-16s max |d bits/dim| default vs direct dispatch %.2e %s" % (k, d, "" if d < 1e-5 else "  <-- ABOVE 1e-5"))
print("worst", worst)
sys.exit(0 if worst < 1e-5 else 1)
