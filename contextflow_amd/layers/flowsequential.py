"""FlowSequential (reference: contextflow/layers/flowsequential.py:5-68).

`forward` = sum of the layers' log-dets + the prior's log-density, shape (B, M).

Two execution modes produce the same numbers:
  * layer-by-layer (any layer list, and always the first call, because ActNorm's data-dependent
    init needs the materialised intermediates);
  * the fused plan used afterwards: pre-processing (Dequantization..LogitTransform[..Augment]) in
    one kernel, each Conv1x1->ActNorm->Coupling triple in ONE fp32-MFMA kernel that accumulates its
    log-det straight into a running per-sample buffer, SplitPrior/prior GMMs accumulated into a
    running (B, M) buffer, channel splits passed by stride (no copies).
Set `fused = False` on the instance to force the layer-by-layer mode."""
import math

import torch
import torch.nn as nn

from . import _hip
from .actnorm import ActNorm
from .augment import Augment
from .conv1x1 import Conv1x1
from .coupling import Coupling
from .dequantize import Dequantization
from .distributions.gaussian import GaussianMixtureDistribution, StandardNormal, gmm_logprob
from .distributions.uniform import UniformDistribution
from .normalize import Normalization
from .splitprior import SplitPrior
from .squeeze import Squeeze, squeeze_op
from .transforms import LogitTransform


class FlowSequential(nn.Module):
    def __init__(self, dist, *modules):
        super().__init__()
        self.dist = dist
        self.mixtures = dist.M
        for i, module in enumerate(modules):
            self.add_module(str(i), module)
        self.sequence_modules = modules
        self.fused = True
        self.step_events = None      # bench.py: list collecting (start, end, batch) HIP events per step-kernel launch

    def __iter__(self):
        yield from self.sequence_modules

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
        for m in self.sequence_modules:
            if isinstance(m, ActNorm) and not m.is_initialized():
                return False
        return True

    @staticmethod
    def _step_supported(conv, act, cpl, shape):
        if not (isinstance(conv, Conv1x1) and isinstance(act, ActNorm) and type(cpl) is Coupling):
            return False
        C, H, W = shape
        k = cpl.NN[2]
        if tuple(k.kernel_size) != (3, 3) or tuple(k.padding) != (1, 1) or conv.D != C or act.D != C:
            return False
        return bool(_hip.lib().cf_flow_step_supported(C, H, W, 3, 3))

    def _forward_fused(self, x, context):
        mods = self.sequence_modules
        n = len(mods)
        B, M, dev = x.shape[0], self.mixtures, x.device
        ld1 = torch.zeros(B, device=dev, dtype=torch.float32)       # per-sample scalar log-dets
        ldM = torch.zeros(B, M, device=dev, dtype=torch.float32)    # per-mixture terms (priors)
        st = _hip.stream()
        i = 0
        while i < n:
            m = mods[i]
            # ---- Dequantization -> Normalization -> Normalization -> LogitTransform [-> Augment]
            if (isinstance(m, Dequantization) and isinstance(m.dist, UniformDistribution) and i + 3 < n
                    and isinstance(mods[i + 1], Normalization) and isinstance(mods[i + 2], Normalization)
                    and isinstance(mods[i + 3], LogitTransform)):
                n1, n2 = mods[i + 1], mods[i + 2]
                xin = _hip.f32(x)
                C, H, W = xin.shape[1:]
                N = C * H * W
                u, _ = m.dist.sample(B, context=xin)
                u = _hip.f32(u)
                aug = mods[i + 4] if (i + 4 < n and isinstance(mods[i + 4], Augment)
                                      and isinstance(mods[i + 4].distribution, StandardNormal)
                                      and mods[i + 4].split_dim == 1) else None
                ca = aug.aug_size if aug is not None else 0
                y = torch.empty(B, C + ca, H, W, device=dev, dtype=torch.float32)
                cst = -N * math.log(n1._s) - N * math.log(n2._s)      # normalize.py:42-49, twice
                ldp = ld1 if i == 0 else torch.empty_like(ld1)           # the kernel assigns its per-sample ldj
                _hip.call("cf_preprocess_fwd", _hip.p(xin), _hip.p(u), _hip.p(y), _hip.p(ldp), B, N, (C + ca) * H * W,
                          n1._t, n1._s, n2._t, n2._s, cst, st)
                if ldp is not ld1:
                    ld1 += ldp
                if aug is not None:
                    eps, logq = aug.distribution.sample(B)
                    y[:, C:].copy_(eps)
                    ld1 -= logq.squeeze(-1)                            # Augment ldj = -log q(eps)
                    i += 5
                else:
                    i += 4
                x = y
                continue
            # ---- Conv1x1 -> ActNorm -> Coupling in one MFMA kernel
            if i + 2 < n and x.dim() == 4 and self._step_supported(m, mods[i + 1], mods[i + 2], tuple(x.shape[1:])):
                x = self._fused_step(x, m, mods[i + 1], mods[i + 2], ld1, st, self.step_events)
                i += 3
                continue
            # ---- Squeeze((2,2)) directly in front of a fused step: folded into the step's operand addressing
            if (isinstance(m, Squeeze) and tuple(m.p) == (2, 2) and i + 3 < n and x.dim() == 4
                    and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                    and self._step_supported(mods[i + 1], mods[i + 2], mods[i + 3],
                                             (x.shape[1] * 4, x.shape[2] // 2, x.shape[3] // 2))):
                x = self._fused_step(x, mods[i + 1], mods[i + 2], mods[i + 3], ld1, st, self.step_events, squeeze=True)
                i += 4
                continue
            if isinstance(m, Squeeze):
                x = squeeze_op(x, m.p, False)
                i += 1
                continue
            if isinstance(m, SplitPrior) and isinstance(m.dist, GaussianMixtureDistribution):
                c = x.shape[1] // 2
                gmm_logprob(x[:, c:], m.dist.prepared(), out=ldM, accumulate=True)
                x = x[:, :c]
                i += 1
                continue
            # ---- anything else: the layer's own kernels
            x, ldj = m(x, context)
            if ldj.dim() == 2:
                ldM += ldj
            else:
                ld1 += ldj
            i += 1
        gmm_logprob(x, self.dist.prepared(), out=ldM, accumulate=True)
        logp = torch.empty(B, M, device=dev, dtype=torch.float32)
        _hip.call("cf_logdet_combine", _hip.p(ldM), _hip.p(ld1), _hip.p(logp), B, M, st)
        return x, logp

    @staticmethod
    def _fused_step(x, conv, act, cpl, ld1, st, events=None, squeeze=False):
        x, xbs = _hip.bview(x)
        B = x.shape[0]
        C, H, W = (x.shape[1] * 4, x.shape[2] // 2, x.shape[3] // 2) if squeeze else tuple(x.shape[1:])
        L = _hip.lib()
        ws = torch.empty(L.cf_flow_step_ws_bytes(C, H, W), device=x.device, dtype=torch.uint8)
        f, pp = _hip.f32, _hip.p
        c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
        _hip.call("cf_flow_step_prepare", pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
                  pp(f(c1.weight.detach())), pp(f(c1.bias.detach())), pp(f(c2.weight.detach())), pp(f(c2.bias.detach())),
                  pp(f(c3.weight.detach())), pp(f(c3.bias.detach())), pp(ws), C, H, W, st)
        z = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
        if events is not None:       # HIP events on the launch stream, bracketing exactly this kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _hip.call("cf_flow_step_fwd", pp(x), pp(z), pp(ld1), pp(ws), B, C, H, W, xbs, int(squeeze), st)
        if events is not None:
            e1.record()
            events.append((e0, e1, B, C))
        return z

    # ------------------------------------------------------------------ reference API
    def forward(self, input, context=None):
        _hip.require_device(input)
        with torch.no_grad():
            if self._fusable():
                return self._forward_fused(input, context)
            return self._forward_layers(input, context)

    def log_prob(self, input, context=None):
        return self.forward(input, context)[1]

    def sample(self, n_samples, context=None):
        z, _ = self.dist.sample(n_samples, context)
        for module in reversed(self.sequence_modules):
            z = module.reverse(z, context)
        return z


class FlowInvSequential(nn.Module):
    """Sampling-direction flow used only by the specialist context encoders (flowsequential.py:42-68)."""

    def __init__(self, dist, *modules):
        super().__init__()
        raise NotImplementedError("FlowInvSequential belongs to the specialist path (SURVEY.md §8(f) rank 2)")
