#!/usr/bin/env python3
"""Launch census of one captured training step (rocprofv3 --kernel-trace): kernels by name with counts and time.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 launch_census.py run <name> <B>;  launch_census.py sum DIR"""
import csv, glob, os, re, sys, collections
if sys.argv[1] == "sum":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    marks = [i for i, r in enumerate(rows) if "FusedAdam" in r[2] or "fused_adam" in r[2].lower() or "k_adamw" in r[2]]
    # one step = between the first fused-Adam launch of consecutive steps
    firsts = [m for j, m in enumerate(marks) if j == 0 or marks[j] - marks[j - 1] > 8]
    a, b = firsts[-2], firsts[-1]
    win = rows[a:b]
    c = collections.defaultdict(lambda: [0, 0])
    for s, e, n in win:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.match(r"([\w:]+(<[^(]{0,40})?)", n).group(1)
        c[k][0] += 1; c[k][1] += e - s
    print("launches / step: %d, span %.2f ms, kernel time %.2f ms" % (len(win), (rows[b][0] - rows[a][0]) / 1e6, sum(v[1] for v in c.values()) / 1e6))
    for k, (n, t) in sorted(c.items(), key=lambda kv: -kv[1][0])[:40]:
        print("%4d  %8.1f us  %s" % (n, t / 1e3, k))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
name, B = sys.argv[2], int(sys.argv[3])
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.rand(B, *ds, device=dev) if M == 1 else torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, max(M, 2), (B,), device=dev) % M
with torch.no_grad():
    model(x)
inv = 1.0 / (ds[0] * ds[1] * ds[2])
loss_fn = (lambda lp, y: -(lp * inv).mean()) if M == 1 else (lambda lp, y: torch.nn.functional.cross_entropy(lp * inv, y))
opt = (cfa.optim.FusedAdamW(model.parameters(), lr=1e-4) if os.environ.get("CF_OWN_ADAMW") == "1"
       else torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True))
step = model.capture_train_step(x, loss_fn, opt)
for _ in range(8):
    step(x, gt)
torch.cuda.synchronize()
