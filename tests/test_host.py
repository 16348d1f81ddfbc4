"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host-side
mirror of the reference API has the reference's state_dict layout, the product refuses to run on
the CPU (no silent fallback), and the multi-rank NLL reduction is correct (gloo, world size 2)."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from contextflow_amd import build
    return build.build()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "contextflow_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    import ctypes
    from contextflow_amd.layers import _hip
    syms = header_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(built)
    for s in syms:
        assert hasattr(lib, s), "header declares %s but the library does not export it" % s
    assert sorted(_hip.SIGNATURES) == syms, set(_hip.SIGNATURES) ^ set(syms)
    assert _hip.lib().cf_abi_version() == _hip.ABI_VERSION == 13
    # pure host-side queries work without a GPU
    assert _hip.lib().cf_flow_step_supported(64, 4, 4, 3, 3) == 1
    assert _hip.lib().cf_flow_step_supported(26, 8, 1, 3, 1) == 0
    assert _hip.lib().cf_flow_step_ws_bytes(16, 16, 16) > 4 * 10064
    assert _hip.lib().cf_actnorm_stats_ws_bytes(16) == 16 * 65 * 2 * 8


def test_host_side_size_queries_of_round_3(built):
    """Host-only entry points (no GPU call): executed multiply-adds of the step kernels as dispatched, the plane / partial
    buffer sizes of the transformer step backward (the Python side slices the plane buffer by the layout of the header),
    the grouped weight gradient's workspace."""
    import ctypes
    from contextflow_amd.layers import _hip
    L = _hip.lib()
    direct = lambda C, H: 40 * C * C * H * H
    # evaluation forward: Winograd at 16x16 always, at 8x8 / 4x4 from 1024 / 2048 samples; backward: direct; weight gradients: Winograd
    assert L.cf_flow_step_macs(256, 16, 16, 16, 0) == direct(16, 16) // 2
    assert L.cf_flow_step_macs(256, 32, 8, 8, 0) == direct(32, 8) and L.cf_flow_step_macs(1024, 32, 8, 8, 0) == direct(32, 8) // 2
    assert L.cf_flow_step_macs(2047, 64, 4, 4, 1) == direct(64, 4) and L.cf_flow_step_macs(2048, 64, 4, 4, 1) == direct(64, 4) // 2
    assert L.cf_flow_step_macs(256, 8, 16, 16, 0) == direct(8, 16) // 2 and L.cf_flow_step_macs(256, 8, 16, 16, 1) == direct(8, 16)
    assert L.cf_flow_step_macs(4096, 16, 16, 16, 2) == direct(16, 16)
    assert L.cf_step_wgrads_macs(4096, 32, 8, 8) == direct(32, 8) // 2
    assert L.cf_flow_step_macs(4096, 26, 8, 1, 0) == 0                     # not a conv-step shape
    # transformer step backward: B = 10 -> Bp = 12: planes [x^T, g_y (8 Bp, 26)] [u0 (4 Bp, 26), g_e (4 Bp, 52)] + depth x 4 Bp x 568
    B, C, depth = 10, 26, 6
    Bp = 12
    assert L.cf_vit_step_bwd_plane_floats(B, C, depth) == 2 * 8 * Bp * 26 + 4 * Bp * (26 + 52) + depth * 4 * Bp * 568
    assert L.cf_vit_step_bwd_ln_floats(B, C, depth) == 3 * (64 + 128 + depth * 256 + 128)
    assert L.cf_vit_step_bwd_ws_bytes(C, depth) > 4 * depth * (192 * 52 + 52 * 64 + 2 * 52 * 52)
    assert L.cf_vit_step_rs_supported(26, 8, 1, 2, 1, 52, 64, 1) == 1 and L.cf_vit_step_rs_supported(38, 144, 1, 2, 1, 76, 64, 1) == 0
    arr = lambda v: (ctypes.c_int * len(v))(*v)
    # grouped weight gradients: partials [N][K] | [N] per row range; about 4096 waves per group, >= 64 rows per range
    one = L.cf_linear_wgrad_group_ws_bytes(arr([4096]), arr([52]), arr([192]), 1)
    assert one == 64 * (192 * 52 + 192) * 4                                # 4096 rows in ranges of 64
    many = L.cf_linear_wgrad_group_ws_bytes(arr([131072] * 26), arr([52] * 26), arr([192] * 26), 26)
    assert many == 26 * 79 * (192 * 52 + 192) * 4                          # 52 tile groups -> 79 ranges (of 1664 rows) per member
    assert L.cf_linear_wgrad_group_ws_bytes(arr([1]), arr([1]), arr([1]), 0) == -1


def test_gfx950_code_object(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", built], capture_output=True, text=True)
    blob = out.stdout + out.stderr
    assert "gfx950" in blob, blob[:500]


@pytest.mark.parametrize("name", ["mnist", "cifar10", "smap"])
def test_state_dict_matches_reference_layout(name):
    """create_model reproduces the reference's state_dict names/shapes (oracle.params.param_spec is
    itself asserted equal to the real reference state_dict in tests/golden/make_golden.py)."""
    import contextflow_amd as cfa
    from oracle import flow_oracle as fo, params as op
    cfg, data_size, M = cfa.preset_config(name)
    model = cfa.create_model(cfg, data_size, M)
    ops, prior, M2 = fo.program(name)
    spec = op.param_spec(ops, prior, M2)
    sd = model.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]), k
    model.load_state_dict(op.gen_params(spec, 0), strict=True)
    n = sum(p.numel() for p in model.parameters())
    assert n == {"mnist": 415360, "cifar10": 1518896, "smap": 936008}[name]


def test_layer_types_follow_program():
    import contextflow_amd as cfa
    from oracle import flow_oracle as fo
    kinds = {"dequant": "Dequantization", "affine": "Normalization", "logit": "LogitTransform", "augment": "Augment",
             "squeeze": "Squeeze", "conv1x1": "Conv1x1", "actnorm": "ActNorm", "coupling": "Coupling",
             "transcoupling": "TransCoupling", "split": "SplitPrior"}
    for name in ("mnist", "cifar10", "smap"):
        cfg, data_size, M = cfa.preset_config(name)
        model = cfa.create_model(cfg, data_size, M)
        ops, _, _ = fo.program(name)
        assert [type(m).__name__ for m in model.sequence_modules] == [kinds[o[0]] for o in ops]


def test_no_cpu_fallback():
    """The product path must fail loudly off-device; the oracle is never imported by the package."""
    import contextflow_amd as cfa
    cfg, data_size, M = cfa.preset_config("mnist")
    model = cfa.create_model(cfg, data_size, M)
    with pytest.raises(RuntimeError, match="ROCm device"):
        model(torch.zeros(2, 1, 32, 32))
    with pytest.raises(RuntimeError, match="ROCm device"):
        cfa.layers.Coupling(8, (3, 3), (1, 1))(torch.zeros(1, 8, 4, 4))
    for root, _, files in os.walk(os.path.join(ROOT, "contextflow_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_unbuilt_names_raise_and_specialist_layout():
    """Names outside the built scope raise at construction; the specialist models that ARE built (eye | onehot +
    uniform context encoders, conv couplings) have the reference's state_dict layout (oracle.params.param_spec is
    checked key for key against the reference by tests/golden/make_golden_specialist.py)."""
    import contextflow_amd as cfa
    from oracle import flow_oracle as fo, params as op
    from tests.helpers import SPECIALIST
    L = cfa.layers
    with pytest.raises(NotImplementedError):
        L.MaskedCoupling(4, context_net=object())
    assert set(L.MaskedCoupling(4, (3, 3), (1, 1)).state_dict()) == {
        "NN.conv%d.%s" % (i, k) for i in (1, 2, 3) for k in ("weight", "bias", "mask")}
    with pytest.raises(NotImplementedError):
        L.ContextEncoder([64], "eye", "vardeq", (16,))          # odd code width: the reference's Augment path
    assert set(L.SplineActivation((2, 2, 2), individual_weights=True).state_dict()) == {
        "unnormalized_widths", "unnormalized_heights", "unnormalized_derivatives"}
    for fxname, (name, ctx) in SPECIALIST.items():
        cfg, ds, M = cfa.preset_config(name)
        cfg.update(generalist=False, enc_emb=ctx["enc_emb"], enc_type=ctx.get("enc_type", "uniform"), contextflow=ctx["contextflow"])
        sd = cfa.create_model(cfg, ds, M, contexts=ctx["contexts"]).state_dict()
        ops, ps, MM = fo.program(name)
        spec = op.param_spec(ops, ps, MM, ctx)
        assert list(sd.keys()) == list(spec.keys()), fxname
        assert all(tuple(sd[k].shape) == tuple(spec[k][0]) for k in sd), fxname


def test_specialist_front_end_takes_uniform_encoders_only():
    """layers/specialist.py::_front_end forms a context code inside cf_linear_group only for encoders whose code IS the uniform
    dequantisation of the one-hot code / of the context itself (model.py:32-49, dequantize.py:26-70); flow-type encoders (vardeq,
    argmax, probsample: the noise comes from a conditional flow) keep their own path.  Host logic only - no kernel runs here."""
    import contextflow_amd as cfa
    from contextflow_amd.layers import specialist
    from contextflow_amd.layers.context import OneHotEncoder, EyeEncoder
    cfg, ds, M = cfa.preset_config("mnist")
    for emb, typ, want in (("onehot", "uniform", 1), ("eye", "uniform", 0), ("onehot", "vardeq", None), ("eye", "argmax", None)):
        contexts = [4, 6] if typ != "argmax" else [4, 16]
        cfg.update(generalist=False, enc_emb=emb, enc_type=typ, contextflow=True)
        flow = cfa.create_model(cfg, ds, M, contexts=contexts)
        assert specialist.supported(flow)
        nets = [m.context_net for m in flow.sequence_modules if getattr(m, "context_net", None)]
        assert nets
        for net in nets:
            got = specialist._uniform_encoder(net)
            if want is None:
                assert got is None, (emb, typ)
            else:
                enc, card, onehot = got
                assert onehot == want and enc is net[1] and isinstance(net[0], OneHotEncoder if want else EyeEncoder)
                assert (card is None) == (want == 0) and enc.D == (sum(contexts) if want else len(contexts)) and net.contexts == contexts
        # without a context (or with the switch off) the front end has nothing to do
        assert specialist._front_end(flow, None, 4, "cpu") == {}


def test_shard_bounds():
    from contextflow_amd.dist import shard_bounds
    for total in (0, 1, 7, 64, 65536 + 3):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys, math, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from contextflow_amd.dist import init_process_group, shard_bounds, allreduce_nll, mean_bits_per_dim
from oracle import flow_oracle as fo
from tests.helpers import load_e2e, e2e_inputs
rank, _, world = init_process_group("gloo")
ops, _, M, params, fx = load_e2e("mnist")
x, u, eps = e2e_inputs("mnist", fx)
lo, hi = shard_bounds(x.shape[0], rank, world)
_, logp = fo.flow_forward(ops, params, x[lo:hi], u[lo:hi], [e[lo:hi] for e in eps])   # oracle stands in for the GPU kernels
red = allreduce_nll(torch.logsumexp(logp.double(), -1).sum(), hi - lo)
full = torch.logsumexp(torch.from_numpy(fx["logp"]).double(), -1)
assert int(red[1]) == x.shape[0]
assert abs(float(red[0]) - float(full.sum())) < 1e-2 * x.shape[0], (float(red[0]), float(full.sum()))
bpd = mean_bits_per_dim(red, 1024)
ref = float(-(full.mean()) / (1024 * math.log(2)))
assert abs(bpd - ref) < 1e-5, (bpd, ref)
dist.barrier()
if rank == 0: print("OK", bpd)
'''


def test_two_rank_nll_allreduce_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


GRAD_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from contextflow_amd.dist import init_process_group, allreduce_gradients
rank, _, world = init_process_group("gloo")
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
for i, p in enumerate(m.parameters()):
    p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
allreduce_gradients(m, bucket_bytes=64)           # tiny buckets: several messages
for i, p in enumerate(m.parameters()):
    assert torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))), (rank, i, p.grad.flatten()[:3])
dist.barrier()
if rank == 0: print("OK")
'''


def test_two_rank_gradient_allreduce_gloo(tmp_path):
    script = tmp_path / "gworker.py"
    script.write_text(GRAD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


INIT_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from contextflow_amd.dist import init_process_group, shard_bounds, allreduce_actnorm_sums, broadcast_parameters
rank, _, world = init_process_group("gloo")
fx = np.load(os.path.join(sys.argv[1], "tests", "golden", "unit_layers.npz"))
x = torch.from_numpy(fx["actnorm/x"]).double()              # the reference's first (initialising) batch, (5, 7, 3, 4)
lo, hi = shard_bounds(x.shape[0], rank, world)              # ragged shards: 3 + 2 samples
xs = x[lo:hi]
C = x.shape[1]
# what cf_actnorm_sums leaves on each rank (fp64 per-channel sums of ITS shard), here with torch standing in for the kernel
sums = torch.cat([xs.sum((0, 2, 3)), (xs * xs).sum((0, 2, 3)), torch.tensor([float(xs.shape[0] * 12)], dtype=torch.float64)])
allreduce_actnorm_sums(sums)
n = float(sums[2 * C])
assert n == 5 * 12
mean = sums[:C] / n                                           # cf_actnorm_from_sums (actnorm.py:31-33)
var = ((sums[C:2 * C] - sums[:C] * mean) / (n - 1.0)).clamp_min(0)
logs = torch.log(var.sqrt() + 1e-8)
assert torch.allclose(mean.float(), torch.from_numpy(fx["actnorm/sd:NN_t"]), rtol=1e-6, atol=1e-7), rank
assert torch.allclose(logs.float(), torch.from_numpy(fx["actnorm/sd:NN_logs"]), rtol=1e-6, atol=1e-7), rank

# bucketed parameter broadcast: fp32 parameters + an int64 buffer, rank 1 starts from different values
torch.manual_seed(rank)
m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
m[1].num_batches_tracked.fill_(40 + rank)
versions = [p._version for p in m.parameters()]
broadcast_parameters(m, src=0)
torch.manual_seed(0)
ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
for a, b in zip(m.parameters(), ref.parameters()):
    assert torch.equal(a, b), rank
assert int(m[1].num_batches_tracked) == 40
assert all(p._version > v for p, v in zip(m.parameters(), versions))     # parameter-derived caches key on the version
dist.barrier()
if rank == 0: print("OK")
'''


def test_two_rank_sharded_actnorm_init_and_bucketed_broadcast_gloo(tmp_path):
    """rank-sharded ActNorm init (all-reduce of per-channel sum x, sum x^2, count) equals the
    global-batch init of the reference's fixture; parameters travel as one flat message per dtype."""
    script = tmp_path / "iworker.py"
    script.write_text(INIT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


BUCKET_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from contextflow_amd.dist import init_process_group, GradBucket
rank, _, world = init_process_group("gloo")
torch.manual_seed(0)
ps = [torch.nn.Parameter(torch.zeros(*sh)) for sh in ((3, 5), (7,), (2, 2, 3, 3), (1,), (6, 1))]
b = GradBucket([ps[:2], ps[2:4], ps[4:]], torch.device("cpu"))
assert len(b.segments) == 3 and b.flat.numel() == 16 + 8 + 36 + 4 + 8
for i, p in enumerate(ps):                                 # the "gradient kernels" write the views; p.grad IS the view
    b.view(p).fill_(float((rank + 1) * (i + 1)))
    p.grad = b.view(p)
    assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * b.slots[p][0] and p.grad.shape == p.shape
for i in range(len(b.segments)):
    b.reduce(i)                                            # one message per segment, enqueued as soon as it is complete
b.finish()
for i, p in enumerate(ps):
    assert torch.equal(p.grad, torch.full_like(p, 1.5 * (i + 1))), (rank, i, p.grad.flatten()[:3])      # mean over the two ranks
assert b.message_bytes() == [96, 160, 32]
dist.barrier()
if rank == 0: print("OK")
'''


def test_two_rank_gradient_bucket_gloo(tmp_path):
    """dist.GradBucket: p.grad are views of ONE flat tensor, every segment is reduced in place (no cat / copy back), the
    result is the mean over the ranks."""
    script = tmp_path / "bworker.py"
    script.write_text(BUCKET_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


@pytest.mark.parametrize("name", ["cifar10", "mnist", "smap"])
def test_gradient_bucket_follows_the_backward(built, name):
    """layers/autograd.py::_bucket_for on the tape the fused plan of a flow would leave (built on the CPU from the plan:
    records carry modules only): every trainable tensor has exactly one slot, slots appear in the order the backward
    produces the gradients (final prior and last level first), a segment closes at every SplitPrior and once it holds
    1 MB, and `closes` names the record after which each segment is complete."""
    import contextflow_amd as cfa
    from contextflow_amd.layers import autograd as ag
    cfg, ds, M = cfa.preset_config(name)
    flow = cfa.create_model(cfg, ds, M)
    for m in flow.modules():
        if hasattr(m, "initialized"):
            m.initialized.fill_(1)
            m._init_done = True
    plan = flow._build_plan(tuple(ds))
    tape = []
    for op in plan:
        if op[0] == "pre":
            tape.append(("pre",))
        elif op[0] == "step":
            tape.append(("step", None, op[5], op[1], op[2], op[3], op[4], None, None, None, None))
        elif op[0] == "vstep":
            tape.append(("vstep", None, op[1], op[2], op[3], None, None))
        elif op[0] == "split":
            tape.append(("split", None, op[1].dist, None))
        elif op[0] == "squeeze":
            tape.append(("squeeze", tuple(op[1].p)))
        else:
            tape.append(("layer", op[1], None))
    tape.append(("prior", None, flow.dist, None))
    params = [p for p in flow.parameters() if p.requires_grad]
    b = ag._bucket_for(flow, tape, params)
    assert set(b.slots) == set(params) and len(b.slots) == len(params)
    lo_prev = -1
    order = [p for ri in range(len(tape) - 1, -1, -1) for p in ag._record_params(tape[ri])]
    for p in order:                                         # backward order, 16-byte aligned, disjoint
        lo, n = b.slots[p]
        assert lo > lo_prev and lo % 4 == 0 and n == p.numel()
        lo_prev = lo + n - 1
    assert b.segments[0][0] == 0 and all(a[1] == c[0] for a, c in zip(b.segments, b.segments[1:])) and b.segments[-1][1] == b.flat.numel()
    assert b.segment_of(flow.dist.mG) == 0                 # the final prior leads the first message
    nsplit = sum(1 for r in tape if r[0] == "split")
    assert len(b.segments) >= nsplit + 1
    closed = sorted(i for v in b.closes.values() for i in v)
    assert closed == list(range(len(b.segments)))
    for ri, segs in b.closes.items():                       # the closing record owns the last slot of its segments
        last = max(b.slots[p][0] for p in ag._record_params(tape[ri]))
        assert b.segments[segs[-1]][0] <= last < b.segments[segs[-1]][1]
    if name == "cifar10":
        mb = [v / 2 ** 20 for v in b.message_bytes()]
        assert abs(sum(mb) - 1518896 * 4 / 2 ** 20) < 0.01 and max(mb) < 2.6, mb
    assert ag._bucket_for(flow, tape, params) is b          # kept: p.grad keeps viewing the same storage


def test_calls_run_on_the_device_that_owns_the_tensors(built, monkeypatch):
    """`_hip.call` launches on the device the tensor arguments live on, on that device's current stream - the reference
    picks `cuda:N` without torch.cuda.set_device (model.py:170).  Host-side check of the selection logic with stand-in
    tensors (a one-GPU box cannot show a second device): a recording entry point replaces the library."""
    import contextlib
    import ctypes
    import types
    from contextflow_amd.layers import _hip

    def fake(index):
        t = types.SimpleNamespace(device=torch.device("cuda", index))
        ptr = _hip._Ptr(0x1000 * (index + 1))
        ptr._keep = t
        return ptr

    log = []

    class FakeLib:
        def cf_probe(self, *args):
            log.append(("launch", state["current"], [a.value for a in args if isinstance(a, _hip._Stream)]))
            return 0

    state = {"current": 0}

    @contextlib.contextmanager
    def device_ctx(dev):
        prev, state["current"] = state["current"], dev.index
        log.append(("enter", dev.index))
        try:
            yield
        finally:
            state["current"] = prev
            log.append(("exit", dev.index))

    monkeypatch.setattr(_hip, "lib", lambda: FakeLib())
    monkeypatch.setattr(_hip, "_current_device", lambda: state["current"])
    monkeypatch.setattr(_hip, "_device_ctx", device_ctx)
    monkeypatch.setattr(_hip, "_current_stream", lambda dev: types.SimpleNamespace(cuda_stream=0xABC0 + dev.index))
    st0 = _hip._Stream(0xABC0)                       # what `_hip.stream()` returned under the current device 0

    _hip.call("cf_probe", fake(0), fake(0), 7, st0)  # tensors on the current device: no guard, the stream as passed
    assert log == [("launch", 0, [0xABC0])]
    del log[:]
    _hip.call("cf_probe", fake(3), None, fake(3), st0)   # tensors on cuda:3 while cuda:0 is current
    assert log == [("enter", 3), ("launch", 3, [0xABC3]), ("exit", 3)], log
    assert state["current"] == 0
    with pytest.raises(RuntimeError, match="different devices"):
        _hip.call("cf_probe", fake(0), fake(1), st0)
    assert _hip.device_of([3, None, ctypes.c_void_p(5)]) is None      # host-only entry points: nothing to select
    cpu, cuda = torch.zeros(1), types.SimpleNamespace(is_cuda=True, device=torch.device("cuda", 1))
    with pytest.raises(RuntimeError, match="no CPU path"):
        _hip.require_device(cpu)
    with pytest.raises(RuntimeError, match="different devices"):
        _hip.require_device(cuda, types.SimpleNamespace(is_cuda=True, device=torch.device("cuda", 0)))


def test_no_library_gemm_in_the_product():
    """Every contraction of the product path is a hand-written kernel reached through the C ABI: no torch matmul."""
    import glob
    pat = re.compile(r"(\s@\s|torch\.(matmul|mm|bmm|einsum|addmm|baddbmm)\b|F\.linear\b|\.matmul\()")
    offenders = []
    for f in glob.glob(os.path.join(ROOT, "contextflow_amd", "**", "*.py"), recursive=True):
        for i, line in enumerate(open(f), 1):
            code = line.split("#", 1)[0]
            if pat.search(code):
                offenders.append("%s:%d: %s" % (os.path.relpath(f, ROOT), i, line.strip()))
    assert not offenders, "\n".join(offenders)
