"""FlowSequential (reference: contextflow/layers/flowsequential.py:5-68).

`forward` = sum of the layers' log-dets + the prior's log-density, shape (B, M).

Two execution modes produce the same numbers:
  * layer-by-layer (any layer list, and always the first call, because ActNorm's data-dependent
    init needs the materialised intermediates);
  * the fused plan used afterwards: pre-processing (Dequantization..LogitTransform[..Augment]) in
    one kernel, each [Squeeze ->] Conv1x1 -> ActNorm -> Coupling group in ONE fp32-MFMA kernel that
    accumulates its log-det straight into a running per-sample buffer, SplitPrior / prior GMMs
    accumulated into a running (B, M) buffer, channel splits passed by stride (no copies).
    The per-call parameter transforms (log|det W|, ActNorm folding, MFMA-fragment packing, GMM
    1/sigma tables) are tiny one-workgroup kernels: they are enqueued on a side HIP stream at the
    start of the call and joined by events, so they overlap the main stream's kernels.
Set `fused = False` on the instance to force the layer-by-layer mode."""
import math

import torch
import torch.nn as nn

from . import _hip
from .actnorm import ActNorm
from .augment import Augment
from .conv1x1 import Conv1x1
from . import coupling as _cpl
from .coupling import Coupling, TransCoupling
from .dequantize import Dequantization
from .distributions.gaussian import (GaussianMixtureDistribution, StandardNormal, gmm_logprob, gmm_levels_ok,
                                     gmm_logprob_levels, GMM_LEVELS_MAX_BATCH)
from .distributions.uniform import UniformDistribution
from .normalize import Normalization
from .permute_axes import PermuteAxes
from .splitprior import SplitPrior
from .squeeze import Squeeze, squeeze_op
from .transforms import LogitTransform

# training: the forward step kernel tapes y0 / h1 / h2 and the backward kernel loads them (no recompute of the two big
# contractions).  False = the tape is dropped after the forward and rebuilt per step at backward time by the same kernel
# (4.5x less tape memory per step, one more forward pass of kernel time; bitwise the same gradients).
TAPE_PLANES = True


VSTEP_TAPE = True        # transformer steps in training: keep the residual-stream tape (False: the backward kernel recomputes it)

# Parameter OBJECTS can be replaced behind a flow's back (`load_state_dict(assign=True)`, `m.weight = nn.Parameter(...)`,
# parametrizations): the new tensor starts at version 0 again, so the version-counter keys of the evaluation caches would
# match stale packed tables.  Every parameter registration in the process bumps this counter (a global torch hook: one integer
# compare per forward, where walking 580 data_ptr()s would cost more than a replayed SMAP forward); a flow that sees a new
# value drops everything it derived from parameters.  Writes through `param.data` (in-place or by assignment) move neither
# the version counter nor this one: call FlowSequential.invalidate_caches() after them.
_PARAM_GENERATION = [0]


def _on_parameter_registration(module, name, param):
    _PARAM_GENERATION[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_on_parameter_registration)


def step_tape(B, C, H, W, dev):
    """Buffers of one step's training tape (cf_flow_step_fwd_taped): y0 (B, C/2, HW), h1, h2 (B, 2C, HW) - the operands of
    the weight-gradient GEMMs - and the opaque aux buffer (log-scales, second half of the Conv1x1+ActNorm output, ReLU
    masks) that is all the step-backward kernel reads of the forward."""
    planes = [torch.empty(B, r, H * W, device=dev, dtype=torch.float32) for r in (C // 2, 2 * C, 2 * C)]
    planes.append(torch.empty(_hip.lib().cf_flow_step_tape_aux_bytes(B, C, H, W), device=dev, dtype=torch.uint8))
    return tuple(planes)


class FlowSequential(nn.Module):
    def __init__(self, dist, *modules):
        super().__init__()
        self.dist = dist
        self.mixtures = dist.M
        for i, module in enumerate(modules):
            self.add_module(str(i), module)
        self.sequence_modules = modules
        self.fused = True
        self.step_events = None      # bench.py: list collecting (start, end, batch, C, H*W) HIP events per step-kernel launch
        self.inv_events = None       # ... per inverse-step launch (sampling)
        self._plans = {}             # input (C,H,W) -> op list
        self._side = {}              # device index -> side stream for the parameter transforms
        # evaluation (no_grad): the packed step workspaces / GMM tables are kept between calls and rebuilt only when a
        # parameter they derive from changes (keyed on the tensors' version counters; `.to()` / `_apply` drops them)
        self._prep = {}              # (plan key, op index) -> (versions, buffers)
        # launch-bound regime: after AUTO_GRAPH_AFTER identical-shape no_grad calls with unchanged parameters the fused
        # forward is captured into a HIP graph and replayed (batches up to AUTO_GRAPH_MAX_BATCH); set False to disable
        self.auto_graph = True
        self._graphs = {}            # (shape, device) -> [stable calls, versions, GraphedFlow | None]
        self._graph_policy = {}      # (shape, device) -> replaying beat eager launches when it was measured (else: stay eager)
        self._tensors = None         # parameters + buffers, collected once (_versions)
        self._rng_key = 0            # Philox key of the in-kernel noise (rank folded in); the stream position is drawn per call
        # data-parallel training (dist.GradBucket, layers/autograd.py): the backward writes the gradients into one flat
        # bucket that p.grad views and all-reduces it segment by segment while it is still running
        self.data_parallel = False
        self._grad_bucket = None
        self._gen = _PARAM_GENERATION[0]

    def __iter__(self):
        yield from self.sequence_modules

    def __getstate__(self):              # streams / cached plans are per-process runtime state
        d = self.__dict__.copy()
        d["_plans"], d["_side"], d["step_events"], d["inv_events"] = {}, {}, None, None
        d["_prep"], d["_graphs"], d["_graph_policy"], d["_tensors"] = {}, {}, {}, None
        d["_grad_bucket"] = None
        d.pop("_cache_holders", None)
        return d

    def invalidate_caches(self):
        """Drop everything derived from parameter VALUES (packed step workspaces, mixture tables, captured graphs).  Needed
        only after writes that bypass the version counters (`param.data.copy_(...)`); optimizer steps, `load_state_dict`,
        `.to()` are noticed without it."""
        self._prep.clear()
        self._graphs.clear()
        self._graph_policy.clear()                   # the replay-vs-eager verdicts are measured again
        self._tensors = None                         # a Parameter OBJECT may have been replaced (load_state_dict(assign=True), m.w = nn.Parameter(..))
        self.__dict__.pop("_fusable_ok", None)       # (an ActNorm may have been reset)
        self.__dict__.pop("_all_params", None)
        self.__dict__.pop("_spec_ws", None)          # layers/specialist.py: packed coupling tables, log|det NN| of frozen Conv1x1,
        self.__dict__.pop("_spec_lad", None)         # Conv1x1.CN in blocked row order
        self.__dict__.pop("_spec_cnb", None)
        self.__dict__.pop("_inv_ws", None)           # packed tables of the inverse steps (`inverse` / `sample`)
        # the modules that keep parameter-derived state of their own: collected once per module tree (`hasattr` on 636 modules
        # cost 1-4 ms of host time per call - a captured SMAP training step at a batch of 256 takes 2 ms and ends with this call)
        holders = self.__dict__.get("_cache_holders")
        if holders is None:
            holders = self.__dict__["_cache_holders"] = [m for m in self.modules() if isinstance(m, (TransCoupling, GaussianMixtureDistribution, Coupling, Conv1x1))]
        for m in holders:
            d = m.__dict__
            d.pop("_lad_cache", None)                # Conv1x1 with a context net under contextflow: H W log|det NN|
            d.pop("_winv_cache", None)               # Conv1x1.reverse: W^-1
            d.pop("_fused_ws", None)                 # TransCoupling._fused: packed ViT table
            d.pop("_ctx_ws", None)                   # Coupling with a context net: packed step tables (forward / backward)
            d.pop("_ctx_wsb", None)
            if "_tab_cache" in d:
                d["_tab_cache"] = None
            if "_flat_cache" in d:
                d["_flat_cache"] = None
            if "_step_src" in d:
                d["_step_src"] = None

    def _apply(self, fn, *a, **k):         # .to() / .cuda() / .float(): new storages, same version counters
        self._prep, self._graphs, self._plans, self._tensors = {}, {}, {}, None
        self._grad_bucket = None
        self.__dict__.pop("_spec_ws", None)
        self.__dict__.pop("_spec_lad", None)
        self.__dict__.pop("_spec_cnb", None)
        self.__dict__.pop("_inv_ws", None)           # packed tables of the inverse steps (`inverse` / `sample`)
        return super()._apply(fn, *a, **k)

    def _versions(self):
        ts = self._tensors
        if ts is None:                       # the module tree is walked once (580 tensors in ~300 modules for the smap flow: per call
            ts = self._tensors = tuple(self.parameters()) + tuple(self.buffers())    # that walk cost more than the replayed forward)
        return tuple(t._version for t in ts)

    # ------------------------------------------------------------------ layer-by-layer mode
    def _forward_layers(self, input, context):
        M, B = self.mixtures, input.shape[0]
        logdet = torch.zeros((B, M), device=input.device)
        for module in self.sequence_modules:
            input, ldj = module(input, context)
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)      # flowsequential.py:23
        return input, self.dist.log_prob(input, context) + logdet

    # ------------------------------------------------------------------ fused plan
    def _fusable(self):
        if not self.fused or not isinstance(self.dist, GaussianMixtureDistribution):
            return False
        # a positive answer holds until the module tree changes (`_gen`) or the caches are dropped: the walk below - ~50 layers through
        # nn.Module.__getattr__ - costs 90 us of host time, per forward call, and an eager training step is host-bound
        ok = self.__dict__.get("_fusable_ok")
        if ok is not None and ok[0] == self._gen and all(a._init_done for a in ok[1]):      # (a loaded state_dict may reset an ActNorm)
            return True
        if getattr(self.dist, "context_net", None):
            return False                     # specialist models run layer by layer (per-sample parameters)
        for m in self.sequence_modules:
            if getattr(m, "context_net", None) or getattr(getattr(m, "dist", None), "context_net", None):
                return False
            if isinstance(m, ActNorm) and not m.is_initialized():
                return False
        self.__dict__["_fusable_ok"] = (self._gen, [m for m in self.sequence_modules if isinstance(m, ActNorm)])
        return True

    def _trainable_params(self):
        """The parameters with requires_grad, in `parameters()` order; the flat parameter list is kept until the module tree changes."""
        allp = self.__dict__.get("_all_params")
        if allp is None or allp[0] != self._gen:
            allp = self.__dict__["_all_params"] = (self._gen, list(self.parameters()))
        return [p for p in allp[1] if p.requires_grad]

    def _specialist(self):
        return bool(getattr(self.dist, "context_net", None))

    def _needs_only_init(self):
        """True when the fused plan is blocked only by ActNorm layers that have not seen their first batch yet."""
        if not self.fused or not isinstance(self.dist, GaussianMixtureDistribution) or getattr(self.dist, "context_net", None):
            return False
        return not any(getattr(m, "context_net", None) or getattr(getattr(m, "dist", None), "context_net", None)
                       for m in self.sequence_modules)

    @staticmethod
    def _step_supported(conv, act, cpl, shape):
        if not (isinstance(conv, Conv1x1) and isinstance(act, ActNorm) and type(cpl) is Coupling):
            return False
        C, H, W = shape
        k = cpl.NN[2]
        if tuple(k.kernel_size) != (3, 3) or tuple(k.padding) != (1, 1) or conv.D != C or act.D != C:
            return False
        return bool(_hip.lib().cf_flow_step_supported(C, H, W, 3, 3))

    def _build_plan(self, shape):
        """Walk the layer list once per input shape and group it into kernels."""
        mods, n = self.sequence_modules, len(self.sequence_modules)
        ops, i = [], 0
        shape = tuple(shape)

        def is3(sh):
            return sh is not None and len(sh) == 3
        while i < n:
            m = mods[i]
            if (isinstance(m, Dequantization) and isinstance(m.dist, UniformDistribution) and i + 3 < n
                    and isinstance(mods[i + 1], Normalization) and isinstance(mods[i + 2], Normalization)
                    and isinstance(mods[i + 3], LogitTransform) and is3(shape)):
                aug = mods[i + 4] if (i + 4 < n and isinstance(mods[i + 4], Augment)
                                      and isinstance(mods[i + 4].distribution, StandardNormal)
                                      and mods[i + 4].split_dim == 1) else None
                ops.append(("pre", m, mods[i + 1], mods[i + 2], aug))
                if aug is not None:
                    shape = (shape[0] + aug.aug_size,) + shape[1:]
                i += 5 if aug is not None else 4
                continue
            if is3(shape) and i + 2 < n and self._step_supported(m, mods[i + 1], mods[i + 2], shape):
                ops.append(("step", m, mods[i + 1], mods[i + 2], shape, False))
                i += 3
                continue
            if (is3(shape) and i + 2 < n and isinstance(m, Conv1x1) and isinstance(mods[i + 1], ActNorm)
                    and isinstance(mods[i + 2], TransCoupling) and m.D == shape[0] and mods[i + 1].D == shape[0]
                    and not m.context_net and not mods[i + 1].context_net and mods[i + 2].step_supported(shape)):
                ops.append(("vstep", m, mods[i + 1], mods[i + 2], shape))       # transformer-coupling step, one kernel
                i += 3
                continue
            if isinstance(m, Squeeze) and is3(shape):
                sq = (shape[0] * m.p[0] * m.p[1], shape[1] // m.p[0], shape[2] // m.p[1])
                if (tuple(m.p) == (2, 2) and shape[1] % 2 == 0 and shape[2] % 2 == 0 and i + 3 < n
                        and self._step_supported(mods[i + 1], mods[i + 2], mods[i + 3], sq)):
                    ops.append(("step", mods[i + 1], mods[i + 2], mods[i + 3], sq, True))   # Squeeze folded in
                    i += 4
                else:
                    ops.append(("squeeze", m))
                    i += 1
                shape = sq
                continue
            if isinstance(m, SplitPrior) and isinstance(m.dist, GaussianMixtureDistribution) and is3(shape):
                ops.append(("split", m))
                shape = (shape[0] // 2,) + shape[1:]
                i += 1
                continue
            if isinstance(m, Augment) and is3(shape) and m.split_dim == 1:
                shape = (shape[0] + m.aug_size,) + shape[1:]
            elif isinstance(m, PermuteAxes) and is3(shape):
                shape = tuple(shape[a - 1] for a in m.permutation[1:])
            elif isinstance(m, (Squeeze, SplitPrior)):
                shape = None            # unknown geometry from here on: everything else runs layer by layer
            ops.append(("layer", m))
            i += 1
        return ops

    def _noise_nonce(self, dev):
        """Per-call position of the in-kernel noise stream: one 63-bit draw from torch's CUDA generator of `dev`, consumed
        the way torch's own random ops consume it.  So the noise follows `torch.manual_seed` (re-seeding reproduces it,
        as it does for the reference's torch.rand / randn draws: uniform.py:32, gaussian.py:69), the generator moves on
        by one draw per forward whatever the user does with it in between, and under graph capture torch registers the
        generator with the graph: every replay reads a fresh value.  Data-parallel ranks fold their rank into the Philox
        KEY: equal seeds on every rank still give every rank its own dequantisation / Augment noise."""
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        self._rng_key = (0x243F6A8885A308D3 ^ (rank * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        return torch.empty(1, device=dev, dtype=torch.int64).random_()

    def _side_stream(self, dev, k=0):
        s = self._side.get((dev.index, k))
        if s is None:
            s = self._side[(dev.index, k)] = torch.cuda.Stream(device=dev)
        return s

    @staticmethod
    def _prepare_step(conv, act, cpl, shape, dev, winv=None):
        """Packed weight tables of a flow step; winv (C, C): also Wm^-1 from the same factorisation (training)."""
        C, H, W = shape
        ws = torch.empty(_hip.lib().cf_flow_step_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
        f, pp = _hip.f32, _hip.p
        c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
        args = (pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
                pp(f(c1.weight.detach())), pp(f(c1.bias.detach())), pp(f(c2.weight.detach())), pp(f(c2.bias.detach())),
                pp(f(c3.weight.detach())), pp(f(c3.bias.detach())), pp(ws))
        if winv is None:
            _hip.call("cf_flow_step_prepare", *args, C, H, W, _hip.stream())
        else:
            _hip.call("cf_flow_step_prepare_train", *args, pp(winv), C, H, W, _hip.stream())
        return ws

    _chain_cache = {}

    @classmethod
    def _chain_max(cls, C, H, W):
        """largest batch at which cf_flow_step_fwd_chain covers this shape (0: never)"""
        if not cls.CHAIN_STEPS:
            return 0
        v = cls._chain_cache.get((C, H, W))
        if v is None:
            v = cls._chain_cache[(C, H, W)] = int(_hip.lib().cf_flow_step_chain_max_batch(C, H, W))
        return v

    CHAIN_STEPS = True               # False: every flow step its own launch at every batch size (A/B, tests)

    @staticmethod
    def _prepare_steps(steps, shape, dev, train=False):
        """Packed tables of SEVERAL flow steps of one shape - steps: [(conv, act, cpl)] - in one factorisation launch and one
        packing launch (cf_flow_step_prepare_batch).  train: also Wm^-1 per step (d log|det W| / dW) and the transposed
        fragments of the backward kernel (cf_flow_step_bwd_prepare_batch).  Returns [ws] or [(ws, winv, wsb)]."""
        C, H, W = shape
        L = _hip.lib()
        f, n = _hip.f32, len(steps)
        wsz = L.cf_flow_step_ws_bytes(C, H, W)
        ws = [torch.empty(wsz, device=dev, dtype=torch.uint8) for _ in range(n)]
        cols = ([f(s[0].NN.detach()) for s in steps], [f(s[1].NN_t.detach()) for s in steps], [f(s[1].NN_logs.detach()) for s in steps],
                [f(s[2].NN[0].weight.detach()) for s in steps], [f(s[2].NN[0].bias.detach()) for s in steps],
                [f(s[2].NN[2].weight.detach()) for s in steps], [f(s[2].NN[2].bias.detach()) for s in steps],
                [f(s[2].NN[4].weight.detach()) for s in steps], [f(s[2].NN[4].bias.detach()) for s in steps])
        A = _hip.ptr_array
        winv = [torch.empty(C, C, device=dev, dtype=torch.float32) for _ in range(n)] if train else None
        _hip.call("cf_flow_step_prepare_batch", n, *[A(c) for c in cols], A(ws), A(winv) if train else None, C, H, W, _hip.stream())
        if not train:
            return ws
        bsz = L.cf_flow_step_bwd_ws_bytes(C, H, W)
        wsb = [torch.empty(bsz, device=dev, dtype=torch.uint8) for _ in range(n)]
        _hip.call("cf_flow_step_bwd_prepare_batch", n, A(cols[0]), A(cols[2]), A(cols[3]), A(cols[5]), A(cols[7]), A(wsb), C, H, W, _hip.stream())
        return list(zip(ws, winv, wsb))

    def _forward_fused(self, x, context, tape=None):
        B, M, dev = x.shape[0], self.mixtures, x.device
        key = tuple(x.shape[1:])
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._build_plan(key)
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev)

        # ---- parameter transforms: kept from the previous call while the parameters they derive from are unchanged
        # (evaluation); otherwise rebuilt on the side stream, overlapping the main stream's kernels
        prepared = {}
        cache_ok = tape is None
        todo = []
        vkey = {}
        # a capturing stream must not wait on an event recorded outside the capture - and need not: torch.cuda.graph
        # synchronises the device before the capture begins, so cached tables are complete by then
        capturing = torch.cuda.is_current_stream_capturing()
        # tables built DURING a capture live in the graph's private pool and their events belong to the capture: they must not
        # outlive it as cache entries (a later eager call would wait on a captured event and read buffers that only exist
        # after a replay).  GraphedFlow warms the cache up before it captures; a user-side capture with a cold cache simply
        # rebuilds the tables inside its graph.
        store_ok = cache_ok and not capturing
        for k, op in enumerate(plan):
            if op[0] == "step":
                srcs = (op[1].NN, op[2].NN_t, op[2].NN_logs) + tuple(p for c in (op[3].NN[0], op[3].NN[2], op[3].NN[4]) for p in (c.weight, c.bias))
            elif op[0] == "vstep":
                srcs = (op[1].NN, op[2].NN_t, op[2].NN_logs) + op[3].step_sources()
                vkey[k] = op[3].step_variant(B)      # small / saturating batches: two kernels, two fragment layouts
            elif op[0] == "split":
                srcs = (op[1].dist.mG, op[1].dist.sG, op[1].dist.wG)
            else:
                continue
            ver = tuple(t._version for t in srcs) + (dev.index,)
            hit = self._prep.get((key, k, vkey.get(k))) if cache_ok else None
            if hit is not None and hit[0] == ver:
                prepared[k] = (hit[1], None if capturing else hit[2])      # the producer's event stays with the entry: a later call on ANOTHER
            else:                                    # stream is ordered against the side stream that wrote the buffers
                todo.append((k, op, ver))
        pver = tuple(t._version for t in (self.dist.mG, self.dist.sG, self.dist.wG)) + (dev.index,)
        hit = self._prep.get((key, "prior")) if cache_ok else None
        prior, ev_prior = (hit[1], None if capturing else hit[2]) if (hit is not None and hit[0] == pver) else (None, None)
        fresh = set()                # entries built by THIS call (their buffers get a record_stream below)
        if todo or prior is None:
            # (one side stream: spreading the tables of the flow steps over 2 / 4 streams was measured on the captured training
            # step at a batch of 256 - smap 2.10 -> 2.16 ms, cifar10 2.49 -> 2.48 ms: tools/dev/prep_streams_ab.py - and dropped)
            sides = [self._side_stream(dev)]
            sides[0].wait_stream(main)
            side = sides[0]
            with torch.cuda.stream(side):
                # conv steps: the tables of all steps of one shape in ONE factorisation + ONE packing launch (+ one for the
                # backward kernel's fragments when training) - cf_flow_step_prepare_batch
                groups = {}
                for k, op, ver in todo:
                    if op[0] == "step":
                        groups.setdefault(tuple(op[4]), []).append((k, op, ver))
                # (one event per group, recorded right behind its launches: the first level's steps start as soon as THEIR
                # tables exist - the 64-channel factorisation alone takes 50 us - instead of behind every table of the flow)
                done, done_ev = {}, {}
                for shape, items in groups.items():
                    bufs = self._prepare_steps([(it[1][1], it[1][2], it[1][3]) for it in items], shape, dev, train=tape is not None)
                    gev = torch.cuda.Event()
                    gev.record(side)
                    for (k, op, ver), buf in zip(items, bufs):
                        done[k] = buf
                        done_ev[k] = gev
                # transformer steps at small batches: the row-split tables of all steps in one launch triple (training: with
                # Wm^-1 and the backward kernel's tables) - cf_vit_step_rs_prepare_batch
                vitems = [(k, op, ver) for k, op, ver in todo if op[0] == "vstep" and vkey[k] == "rs"]
                if vitems:
                    bufs = TransCoupling.step_prepare_rs_batch([(it[1][3], it[1][1].NN, it[1][2].NN_t, it[1][2].NN_logs) for it in vitems], dev,
                                                               train=tape is not None)
                    gev = torch.cuda.Event()
                    gev.record(side)
                    for (k, op, ver), buf in zip(vitems, bufs):
                        done[k] = buf
                        done_ev[k] = gev
                for k, op, ver in todo:
                    if k in done:
                        buf, ev = done[k], done_ev[k]
                    else:
                        if op[0] == "vstep":
                            buf = op[3].step_prepare(op[1].NN, op[2].NN_t, op[2].NN_logs, dev, vkey[k])
                        else:
                            buf = op[1].dist.prepared()
                        ev = torch.cuda.Event()
                        ev.record(side)
                    prepared[k] = (buf, ev)
                    fresh.add(k)
                    if store_ok:
                        self._prep[(key, k, vkey.get(k))] = (ver, buf, ev)
            if prior is None:
                with torch.cuda.stream(side):
                    prior = self.dist.prepared()
                    ev_prior = torch.cuda.Event()
                    ev_prior.record(side)
                    fresh.add("prior")
                    if store_ok:
                        self._prep[(key, "prior")] = (pver, prior, ev_prior)

        # running log-dets: per-sample scalar terms / per-mixture terms (priors).  The first writer ASSIGNS (no zero-fill
        # launches): the pre-processing kernel for ld1, the first mixture kernel for ldM
        ld1 = (torch.empty if plan and plan[0][0] == "pre" else torch.zeros)(B, device=dev, dtype=torch.float32)
        ldM = torch.empty(B, M, device=dev, dtype=torch.float32)
        ldM_set = False
        levels = [] if tape is None and B <= GMM_LEVELS_MAX_BATCH else None
        st = _hip.stream()
        chained = set()              # plan indices that ran as the tail of a chained launch
        for k, op in enumerate(plan):
            kind = op[0]
            if k in chained:
                continue
            if kind == "pre":
                if tape is not None:
                    tape.append(("pre",))
                _, deq, n1, n2, aug = op
                xin = _hip.f32(x)
                C, H, W = xin.shape[1:]
                N = C * H * W
                ca = aug.aug_size if aug is not None else 0
                y = torch.empty(B, C + ca, H, W, device=dev, dtype=torch.float32)
                cst = -N * math.log(n1._s) - N * math.log(n2._s)      # normalize.py:42-49, twice
                ldp = ld1 if k == 0 else torch.empty_like(ld1)         # the kernel assigns its per-sample ldj
                if (deq.dist.fixed_noise is None and (aug is None or aug.distribution.fixed_noise is None)
                        and N % 4 == 0 and (ca * H * W) % 4 == 0):
                    # noise drawn inside the kernel (Philox, keyed by torch's seed): dequantisation uniforms and the
                    # Augment channel's normals never travel through HBM as tensors of their own
                    nonce = self._noise_nonce(dev)
                    _hip.call("cf_preprocess_rng_fwd", _hip.p(xin), _hip.p(y), _hip.p(ldp), _hip.p(nonce), self._rng_key, B, N,
                              ca * H * W, (C + ca) * H * W, n1._t, n1._s, n2._t, n2._s, cst, 0, st)
                    if ldp is not ld1:
                        ld1 += ldp
                    x = y
                    continue
                u = _hip.f32(deq.dist.sample(B, context=xin)[0])
                _hip.call("cf_preprocess_fwd", _hip.p(xin), _hip.p(u), _hip.p(y), _hip.p(ldp), B, N, (C + ca) * H * W,
                          n1._t, n1._s, n2._t, n2._s, cst, st)
                if ldp is not ld1:
                    ld1 += ldp
                if aug is not None:
                    eps, logq = aug.distribution.sample(B)
                    y[:, C:].copy_(eps)
                    ld1 -= logq.squeeze(-1)                            # Augment ldj = -log q(eps)
                x = y
            elif kind == "step":
                _, conv, act, cpl, (C, H, W), sq = op
                ws, ev = prepared[k]
                if ev is not None:
                    main.wait_event(ev)
                if tape is not None:
                    ws, winv, wsb = ws
                    # training: the taping forward kernel writes the step tape and the backward kernel reads it.  If the
                    # planes of this step would take more than 1/64 of the device memory (huge batches), or with
                    # TAPE_PLANES = False, the tape is NOT kept (4.5x less memory per step): the backward then re-runs this
                    # very kernel from the step input.  The forward is the taping kernel in both cases, so the masks the
                    # backward uses are bit for bit those of the forward that produced the loss.
                    keep = TAPE_PLANES and 18 * B * C * H * W <= torch.cuda.get_device_properties(dev).total_memory // 64
                    planes = step_tape(B, C, H, W, dev)
                    tape.append(("step", x, sq, conv, act, cpl, (C, H, W), ws, winv, planes if keep else None, wsb))
                    x, xbs = _hip.bview(x)
                    z = torch.empty(B, C, H, W, device=dev, dtype=torch.float32)
                    _hip.call("cf_flow_step_fwd_taped", _hip.p(x), _hip.p(z), _hip.p(ld1), _hip.p(ws), _hip.p(planes[0]),
                              _hip.p(planes[1]), _hip.p(planes[2]), _hip.p(planes[3]), B, C, H, W, xbs, int(sq), st)
                    del planes
                    x = z
                    continue
                x, xbs = _hip.bview(x)
                z = torch.empty(B, C, H, W, device=dev, dtype=torch.float32)
                events = self.step_events
                if events is None and 0 < B <= self._chain_max(C, H, W):
                    # small batch: this step and the following steps of the same shape (a resolution level) in ONE launch
                    run = [k]
                    while (len(run) < 4 and run[-1] + 1 < len(plan) and plan[run[-1] + 1][0] == "step"
                           and tuple(plan[run[-1] + 1][4]) == (C, H, W) and not plan[run[-1] + 1][5]):
                        run.append(run[-1] + 1)
                    if len(run) > 1:
                        tabs = [ws]
                        for j in run[1:]:
                            wj, evj = prepared[j]
                            if evj is not None:
                                main.wait_event(evj)
                            tabs.append(wj)
                        _hip.call("cf_flow_step_fwd_chain", _hip.p(x), _hip.p(z), _hip.p(ld1), _hip.ptr_array(tabs), len(run), B, C, H, W, xbs,
                                  int(sq), st)
                        chained.update(run[1:])
                        x = z
                        continue
                if events is not None:       # HIP events on the launch stream, bracketing exactly this kernel
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(main)
                _hip.call("cf_flow_step_fwd", _hip.p(x), _hip.p(z), _hip.p(ld1), _hip.p(ws), B, C, H, W, xbs, int(sq), st)
                if events is not None:
                    e1.record(main)
                    events.append((e0, e1, B, C, H * W))
                x = z
            elif kind == "vstep":
                ws, ev = prepared[k]
                winv_v = wsb_v = None
                if isinstance(ws, tuple):    # training, row-split form: (tables, Wm^-1, backward tables) from the batched prepare
                    ws, winv_v, wsb_v = ws
                if tape is not None:         # training: the same one-kernel forward; the backward re-runs the step from its input
                    #                          (and reuses the packed forward table when it is the row-split one)
                    # the residual stream at the layer boundaries goes to a tape (5.8 KB per sample and step) and the backward
                    # kernel does not run the six layers a second time
                    xt = None
                    if VSTEP_TAPE:
                        depth = len(op[3].NN[0].transformer.layers)
                        xt = torch.empty(_hip.lib().cf_vit_step_tape_floats(B, x.shape[1], depth), device=dev, dtype=torch.float32)
                    tape.append(("vstep", x, op[1], op[2], op[3], ws if vkey[k] == "rs" else None, xt, winv_v, wsb_v))
                    if ev is not None:
                        main.wait_event(ev)
                    x = op[3].step_forward(x, ws, ld1, variant=vkey[k], xtape=xt)
                    continue
                if ev is not None:
                    main.wait_event(ev)
                if self.CHAIN_STEPS and vkey[k] == "rs" and _cpl.VIT_EVENTS is None:
                    # small batch (row-split form): this step and the transformer steps that follow it in ONE launch
                    run = [k]
                    nmax = int(_hip.lib().cf_vit_step_rs_chain_max_steps())
                    depth = len(op[3].NN[0].transformer.layers)
                    while (len(run) < nmax and run[-1] + 1 < len(plan) and plan[run[-1] + 1][0] == "vstep" and vkey.get(run[-1] + 1) == "rs"
                           and len(plan[run[-1] + 1][3].NN[0].transformer.layers) == depth):
                        run.append(run[-1] + 1)
                    if len(run) > 1:
                        tabs = [ws]
                        for j in run[1:]:
                            wj, evj = prepared[j]
                            if evj is not None:
                                main.wait_event(evj)
                            tabs.append(wj)
                        xv, xbs = _hip.bview(x)
                        z = torch.empty(B, xv.shape[1], xv.shape[2], xv.shape[3], device=dev, dtype=torch.float32)
                        _hip.call("cf_vit_step_rs_fwd_chain", _hip.p(xv), _hip.p(z), _hip.p(ld1), _hip.ptr_array(tabs), len(run), B, xv.shape[1],
                                  depth, xbs, st)
                        chained.update(run[1:])
                        x = z
                        continue
                x = op[3].step_forward(x, ws, ld1, variant=vkey[k])
            elif kind == "squeeze":
                if tape is not None:
                    tape.append(("squeeze", tuple(op[1].p)))
                x = squeeze_op(x, op[1].p, False)
            elif kind == "split":
                prep, ev = prepared[k]
                if ev is not None:
                    main.wait_event(ev)
                if tape is not None:
                    tape.append(("split", x, op[1].dist, prep))
                c = x.shape[1] // 2
                if levels is not None:       # small batch: every mixture of the flow in one launch pair at the end
                    levels.append((x[:, c:], prep))
                else:
                    gmm_logprob(x[:, c:], prep, out=ldM, accumulate=ldM_set)
                    ldM_set = True
                x = x[:, :c]
            else:                        # any other layer: its own kernels
                if tape is not None:
                    tape.append(("layer", op[1], x))
                x, ldj = op[1](x, context)
                if ldj.dim() == 2:
                    if ldM_set:
                        ldM += ldj
                    else:
                        ldM.copy_(ldj.expand_as(ldM))
                        ldM_set = True
                else:
                    ld1 += ldj
        if ev_prior is not None:
            main.wait_event(ev_prior)
        if tape is not None:
            tape.append(("prior", x, self.dist, prior))
        if levels is not None and gmm_levels_ok(levels + [(x, prior)]):
            logp = gmm_logprob_levels(levels + [(x, prior)], ldM if ldM_set else None, ld1)
        else:
            for xl, prep in levels or ():
                gmm_logprob(xl, prep, out=ldM, accumulate=ldM_set)
                ldM_set = True
            gmm_logprob(x, prior, out=ldM, accumulate=ldM_set)
            logp = torch.empty(B, M, device=dev, dtype=torch.float32)
            _hip.call("cf_logdet_combine", _hip.p(ldM), _hip.p(ld1), _hip.p(logp), B, M, st)
        # buffers written on the side stream are consumed on the main stream: keep the allocator informed
        def _bufs(t):
            if torch.is_tensor(t):
                yield t
            elif isinstance(t, tuple):
                for u in t:
                    yield from _bufs(u)

        for k, v in prepared.items():
            if k in fresh:
                for buf in _bufs(v[0]):
                    buf.record_stream(main)
        if "prior" in fresh:
            for buf in prior:
                if torch.is_tensor(buf):
                    buf.record_stream(main)
        return x, logp

    # ------------------------------------------------------------------ reference API
    def forward(self, input, context=None):
        _hip.require_device(input)
        if self._gen != _PARAM_GENERATION[0]:            # a Parameter object was (re)registered somewhere: see _PARAM_GENERATION
            self._gen = _PARAM_GENERATION[0]
            self.__dict__.pop("_cache_holders", None)
            self.invalidate_caches()
        if torch.is_grad_enabled() and self._specialist():
            from .autograd_ctx import trainable as _trainable
            params = self._trainable_params()
            if params and _trainable(self):  # specialist training under contextflow (autograd_ctx.py); other
                                             # specialist models evaluate only: they fall through to the no_grad path
                if any(isinstance(m, ActNorm) and m.contextflow and not m.is_initialized() for m in self.sequence_modules):
                    with torch.no_grad():    # first call: the ActNorm data-dependent init
                        self._forward_layers(input, context)
                from .autograd_ctx import SpecialistLogProb
                return SpecialistLogProb.apply(self, input, context, *params)
        if torch.is_grad_enabled():
            params = self._trainable_params()
            if params and not self._fusable() and self._needs_only_init():
                with torch.no_grad():        # first training call: the ActNorm data-dependent init (actnorm.py:28-35)
                    self._forward_layers(input, context)
            if params and self._fusable():   # training: same kernels + a tape, hand-written backward (autograd.py)
                from .autograd import FlowLogProb
                return FlowLogProb.apply(self, input, *params)
        with torch.no_grad():
            if self._fusable():
                g = self._auto_graph(input)
                if g is not None:
                    z, logp = g(input)
                    return z.clone(), logp.clone()       # the graph's static outputs are overwritten by the next replay
                return self._forward_fused(input, context)
            if self.fused and self._specialist():
                from . import specialist
                if specialist.supported(self):       # grouped evaluation plan of the specialist conv flows
                    return specialist.forward_eval(self, input, context)
            return self._forward_layers(input, context)

    AUTO_GRAPH_AFTER = 2
    AUTO_GRAPH_MAX_BATCH = 4096

    def _auto_graph(self, x):
        """Graph replay for small, repeated evaluation batches (the reference's operating point is B = 256, config.py:10:
        ~20 launches of a few microseconds of work each).  Conditions: same shape / dtype / device as the previous calls,
        parameters unchanged since (version counters), in-kernel noise (no injected test noise), no event probes, not
        already capturing.  Anything else runs eagerly, and a parameter update drops the graph."""
        from . import coupling as _cpl
        if (not self.auto_graph or self.step_events is not None or _cpl.VIT_EVENTS is not None or x.dim() != 4 or x.shape[0] == 0
                or x.shape[0] > self.AUTO_GRAPH_MAX_BATCH or torch.cuda.is_current_stream_capturing()):
            return None
        for m in self.sequence_modules:
            d = getattr(m, "dist", None) or getattr(m, "distribution", None)
            if d is not None and getattr(d, "fixed_noise", None) is not None:
                return None
        gkey = (tuple(x.shape), x.dtype, x.device)
        ver = self._versions()
        st = self._graphs.get(gkey)
        if st is None or st[1] != ver:
            if len(self._graphs) >= 8:                   # bounded: each graph owns its intermediates
                self._graphs.clear()
            self._graphs[gkey] = [1, ver, None]
            return None
        if st[2] is None:
            if self._graph_policy.get(gkey) is False:    # measured before for this shape: eager launches are faster
                return None
            st[0] += 1
            if st[0] <= self.AUTO_GRAPH_AFTER:
                return None
            st[2] = GraphedFlow(self, x, warmup=1)
            if gkey not in self._graph_policy:
                self._graph_policy[gkey] = self._replay_wins(st[2], x)
                if not self._graph_policy[gkey]:
                    st[2] = None
                    return None
        return st[2]

    def _replay_wins(self, g, x, n=5, rounds=2, margin=0.97):
        """A replayed node costs ~7-12 us whatever its kernel does, eager launches are queued ahead of the running kernel:
        with kernels longer than that (the transformer steps: ~30 us each at a batch of 256) the eager forward can be the
        faster one.  Measured once per input shape (the answer does not depend on parameter values; invalidate_caches()
        forgets it): n eager forwards against n replays between HIP events on the launch stream, best of `rounds` each -
        device time, so that a busy host core (bench.py's CPU-baseline threads) cannot pin the verdict - and the graph is
        only rejected when eager is faster by more than 3 %.  The generator is handed back as it was, so the noise
        sequence does not see the probe."""
        dev = x.device
        gen = torch.cuda.default_generators[dev.index]
        gstate = gen.get_state()
        stream = torch.cuda.current_stream(dev)
        best = []
        for run in (lambda: self._forward_fused(x, None), lambda: g(x)):
            run()
            t = []
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(n):
                    run()
                e1.record(stream)
                e1.synchronize()
                t.append(e0.elapsed_time(e1))
            best.append(min(t))
        gen.set_state(gstate)
        return best[1] * margin < best[0]

    def capture_train_step(self, example_input, loss_fn, optimizer, warmup=1, data_parallel=None):
        """One whole training step - forward, loss, hand-written backward, optimizer update - captured into ONE HIP graph
        (at the reference's batch of 256 a step is ~330 launches of a few microseconds each: launch-bound).  Returns
        `step(x, *loss_args) -> loss` that copies its arguments into static buffers and replays; `loss_fn(logp, *loss_args)`
        maps the (B, M) log-densities to a scalar.  The optimizer must be capturable (`torch.optim.AdamW(..., capturable=
        True)`); ActNorm layers must be initialised (run one forward first).  The first call runs `warmup` eager steps
        (real updates; default 1) and captures; see GraphedTrainStep.
        data_parallel (default: whether a process group of more than one rank exists): every rank steps on ITS shard of the
        batch and the gradients are averaged over the ranks inside the step - written by the backward kernels into one flat
        bucket that `p.grad` views, all-reduced (RCCL) segment by segment while the backward of the lower levels still runs,
        the optimizer waits for the last one; the collectives are captured with the step (`self.data_parallel`,
        dist.GradBucket)."""
        if data_parallel is None:
            from .. import dist as cdist
            data_parallel = cdist._active()
        self.data_parallel = bool(data_parallel)
        return GraphedTrainStep(self, example_input, loss_fn, optimizer, warmup)

    def log_prob(self, input, context=None):
        return self.forward(input, context)[1]

    def capture(self, example_input):
        """Capture the fused forward for inputs of this exact shape into a HIP graph; returns a callable
        `g(x) -> (z, logp)` (static output buffers, overwritten by the next replay)."""
        return GraphedFlow(self, example_input)

    def _inverse_step(self, z, conv, act, cpl, unsqueeze=False):
        """Conv1x1^-1 o ActNorm^-1 o Coupling^-1 in one MFMA kernel (cf_flow_step_inv); unsqueeze: the Squeeze((2,2)) in front of
        the step is inverted by the kernel's stores."""
        z, zbs = _hip.bview(z)
        B, C, H, W = z.shape
        dev = z.device
        f, pp, st = _hip.f32, _hip.p, _hip.stream()
        # the packed tables of the step (forward fragments of the conditioner, W^-1 and the inverse ActNorm) are kept while the
        # parameters they derive from are unchanged - version counter and storage of every source tensor, as the forward's tables
        # (two factorisations + two packing launches per step and call before: 13 - 17 % of a `sample` call at 16 384 samples)
        srcs = (conv.NN, act.NN_t, act.NN_logs, cpl.NN[0].weight, cpl.NN[0].bias, cpl.NN[2].weight, cpl.NN[2].bias, cpl.NN[4].weight,
                cpl.NN[4].bias)
        ver = tuple((t._version, t.data_ptr()) for t in srcs) + (C, H, W, str(dev))
        cache = self.__dict__.setdefault("_inv_ws", {})
        hit = cache.get(id(cpl))
        capturing = torch.cuda.is_current_stream_capturing()
        if hit is not None and hit[0] == ver and not capturing:
            ws, wsi = hit[1], hit[2]
            torch.cuda.current_stream(dev).wait_event(hit[3])      # (another stream than the one that packed them: ordered behind it)
        else:
            ws = self._prepare_step(conv, act, cpl, (C, H, W), dev)
            wsi = torch.empty(_hip.lib().cf_flow_step_inv_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
            _hip.call("cf_flow_step_inv_prepare", pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
                      pp(wsi), C, H, W, st)
            if not capturing:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                cache[id(cpl)] = (ver, ws, wsi, ev)
        x = torch.empty((B, C // 4, 2 * H, 2 * W) if unsqueeze else (B, C, H, W), device=dev, dtype=torch.float32)
        events = self.inv_events
        if events is not None:               # HIP events on the launch stream, bracketing exactly this kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        _hip.call("cf_flow_step_inv", pp(z), pp(x), pp(ws), pp(wsi), B, C, H, W, zbs, int(unsqueeze), st)
        if events is not None:
            e1.record(torch.cuda.current_stream(dev))
            events.append((e0, e1, B, C, H * W))
        return x

    def inverse(self, z, context=None):
        """Run every layer's `reverse` from the last to the first (the loop of flowsequential.py:35-37); groups
        Coupling <- ActNorm <- Conv1x1 of a supported shape run as one fused kernel."""
        _hip.require_device(z)
        mods = self.sequence_modules
        i = len(mods) - 1
        with torch.no_grad():
            while i >= 0:
                m = mods[i]
                if (self.fused and i >= 2 and z.dim() == 4 and isinstance(mods[i - 1], ActNorm)
                        and mods[i - 1].is_initialized()
                        and self._step_supported(mods[i - 2], mods[i - 1], m, tuple(z.shape[1:]))):
                    sq = i >= 3 and isinstance(mods[i - 3], Squeeze) and tuple(mods[i - 3].p) == (2, 2)
                    z = self._inverse_step(z, mods[i - 2], mods[i - 1], m, unsqueeze=sq)
                    i -= 4 if sq else 3
                elif self.fused and self._tail_fusable(i, z):
                    z = self._inverse_tail(z, i)
                    i = -1
                else:
                    z = m.reverse(z, context)
                    i -= 1
        return z

    def _tail_fusable(self, i, z):
        """mods[:i + 1] is the pre-processing of the image flows - Dequantization, Normalization x 2, LogitTransform [, Augment over
        the channels] - whose reverse chain cf_postprocess_inv runs in one pass."""
        mods = self.sequence_modules
        if z.dim() != 4 or i not in (3, 4) or not (isinstance(mods[0], Dequantization) and isinstance(mods[1], Normalization)
                                                   and isinstance(mods[2], Normalization) and isinstance(mods[3], LogitTransform)):
            return False
        return i == 3 or (isinstance(mods[4], Augment) and mods[4].split_dim == 1 and 0 < mods[4].aug_size < z.shape[1])

    def _inverse_tail(self, z, i):
        mods = self.sequence_modules
        z, zbs = _hip.bview(z)
        B, C, H, W = z.shape
        keep = C - (mods[4].aug_size if i == 4 else 0)
        x = torch.empty(B, keep, H, W, device=z.device, dtype=torch.float32)
        n1, n2 = mods[1], mods[2]
        _hip.call("cf_postprocess_inv", _hip.p(z), _hip.p(x), B, keep * H * W, zbs, n2._t, n2._s, n1._t, n1._s, _hip.stream())
        return x

    def sample(self, n_samples, context=None):
        z = self.dist.sample(n_samples, context, need_log_prob=False)[0] if isinstance(self.dist, GaussianMixtureDistribution) \
            else self.dist.sample(n_samples, context)[0]           # (the reference's sample() also returns log p(z): not needed here)
        return self.inverse(z, context)


class GraphedFlow:
    """A FlowSequential forward captured once into a HIP graph (launch-bound regime: small batches, where the ~45
    kernel launches + side-stream hand-offs of a call cost more than the kernels).  Replays write into static
    output buffers; the noise layers keep drawing fresh noise (graph-safe Philox offsets)."""

    def __init__(self, flow, example, warmup=2):
        _hip.require_device(example)
        if not flow._fusable():
            raise RuntimeError("capture needs initialised ActNorms (run one forward first) and the fused plan")
        self.flow = flow
        self.static_in = example.detach().clone()
        with torch.no_grad():
            s = torch.cuda.Stream(device=example.device)
            s.wait_stream(torch.cuda.current_stream(example.device))
            gen = torch.cuda.default_generators[example.device.index]
            gstate = gen.get_state()                   # the warm-up passes draw noise positions nobody sees: hand the
            with torch.cuda.stream(s):                 # generator back as it was, so that capturing is invisible in the
                for _ in range(warmup):               # noise sequence (same seed => same noise, captured or not)
                    flow._forward_fused(self.static_in, None)       # warm-up off the default stream: first-use attribute
            torch.cuda.current_stream(example.device).wait_stream(s)   # calls, allocator pools, plan construction
            gen.set_state(gstate)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_z, self.static_logp = flow._forward_fused(self.static_in, None)

    def __call__(self, x):
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_z, self.static_logp


class GraphedTrainStep:
    """See FlowSequential.capture_train_step.

    First call: `warmup` (default 1) EAGER training steps on the given batch - real optimizer updates, off the default
    stream - then the capture of one more step (a capture records, it does not execute: no update).  It returns the loss
    of the last eager step, so with the default warmup the first call is exactly one update, like every later call.
    Later calls replay.  A replay changes the parameters on the device without moving their version counters, which
    the evaluation caches of the flow key on (packed step tables, mixture tables, auto-captured forward graphs): every
    call therefore ends with `flow.invalidate_caches()`, and a `log_prob` between training steps sees the current
    parameters."""

    def __init__(self, flow, example, loss_fn, optimizer, warmup=1, loss_args=()):
        _hip.require_device(example)
        if not flow._fusable() and not flow._specialist():
            raise RuntimeError("capture_train_step needs initialised ActNorms: run one forward first")
        if warmup < 1:
            raise ValueError("capture_train_step: warmup must be >= 1 (the first call returns the loss of its last eager step)")
        self.flow, self.loss_fn, self.opt = flow, loss_fn, optimizer
        self.static_in = example.detach().clone()
        self.static_args = None
        self.graph = None
        self.warmup = warmup
        self.updates = 0             # optimizer updates performed through this object

    @staticmethod
    def _drain_collectives(dev):
        """Before a capture that contains RCCL collectives: let the process group's watchdog thread retire the EAGER
        collectives of the warm-up step.  It polls their completion events (hipEventQuery) every 100 ms; once the capture has
        pulled RCCL's internal stream into capture mode, HIP answers such a query with hipErrorCapturedEvent ("event last
        recorded in a capturing stream" - for an event recorded BEFORE the capture, on a stream that is capturing NOW), and
        the watchdog aborts the process.  Seen once in four captures with a warm-up step right in front.  The device is idle
        after the synchronize, so every pending work is complete and the next poll (or the one after) drops it."""
        import time
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized() and tdist.get_backend() == "nccl":
            torch.cuda.synchronize(dev)
            time.sleep(0.35)

    def _step(self):
        # grads start as None: the autograd engine then TAKES the gradient buffers the backward returns (allocated from
        # the graph's private pool during capture, so their addresses are the ones every replay writes and the
        # optimizer's captured kernels read) instead of a fill + an accumulate launch per parameter
        self.opt.zero_grad(set_to_none=True)
        logp = self.flow.log_prob(self.static_in)
        loss = self.loss_fn(logp, *self.static_args)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self, x, *loss_args):
        if self.graph is None:
            self.static_args = tuple(a.detach().clone() if torch.is_tensor(a) else a for a in loss_args)
            self.static_in.copy_(x)
            dev = self.static_in.device
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):                   # warm-up steps are real optimizer steps
                for _ in range(self.warmup):
                    first = self._step().detach().clone()
            torch.cuda.current_stream(dev).wait_stream(s)
            self.updates += self.warmup
            if self.flow.data_parallel:
                self._drain_collectives(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_loss = self._step()
            self.flow.invalidate_caches()
            return first
        self.static_in.copy_(x)
        for dst, src in zip(self.static_args, loss_args):
            if torch.is_tensor(dst):
                dst.copy_(src)
        self.graph.replay()
        self.updates += 1
        self.flow.invalidate_caches()
        return self.static_loss


class FlowInvSequential(nn.Module):
    """Sampling-direction flow of the variational context encoders (flowsequential.py:42-68): draw from `dist`, push
    the sample through the layers' forward, subtract their log-dets from the log-density."""

    def __init__(self, dist, *modules):
        super().__init__()
        self.dist = dist
        for i, module in enumerate(modules):
            self.add_module(str(i), module)
        self.sequence_modules = modules

    def __iter__(self):
        yield from self.sequence_modules

    def forward(self, input, context=None):
        return self.sample(input, context)

    def log_prob(self, input, context=None):
        raise RuntimeError("InverseFlow does not support log_prob, see Flow instead.")

    def sample(self, input, context=None):
        with torch.no_grad():
            output, logprob = self.dist.sample(input.size(0), context)
            for module in self.sequence_modules:
                output, ldj = module(output, context)
                logprob = logprob - ldj
        return output, logprob
