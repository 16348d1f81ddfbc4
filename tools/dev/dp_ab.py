import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
import bench
os.environ["CF_DIST_SINGLE_RANK"] = "1"; os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29577"
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1)
for name, B in (("smap", 256), ("smap", 32768), ("cifar10", 256)):
    for dp in (False, True):
        model, cfg = bench.build(name, dev)
        x = bench.synth(name, B, dev, seed=4000)
        gt = torch.randint(0, model.mixtures, (B,), device=dev)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
        step = model.capture_train_step(x, bench.reference_loss(name), opt, data_parallel=dp)
        for _ in range(5): step(x, gt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50 if B <= 1024 else 10
        for _ in range(n): step(x, gt)
        torch.cuda.synchronize()
        print("%s B=%d data_parallel=%s: %.3f ms per captured step" % (name, B, dp, (time.perf_counter() - t0) / n * 1e3), flush=True)
        del model, opt, step
dist.destroy_process_group()
