"""ActNorm (reference: contextflow/layers/actnorm.py:7-101), context-free branch.

Reference quirks kept on purpose: the data-dependent init runs on the
first forward in any mode, uses the unbiased std and log(std + 1e-8); ldj = +sum_c logs."""
import torch
import torch.nn as nn

from . import _hip
from .flowlayer import FlowLayer, encoder_noise


class ActNorm(FlowLayer):
    def __init__(self, data_size, context_net=None, contextflow=False):
        super().__init__()
        D, H, W = data_size if len(data_size) == 3 else (data_size[0], 1, 1)
        self.D, self.H, self.W = D, H, W
        self.NN_t = nn.Parameter(torch.zeros(D))
        self.NN_logs = nn.Parameter(torch.zeros(D))
        self.register_buffer("initialized", torch.tensor(0))
        self.context_net = context_net
        self.contextflow = contextflow
        if self.context_net:                                # actnorm.py:19-26
            self.C = self.context_net.C
            self.CN = nn.Linear(self.C, 2 * D)
            if self.contextflow:
                self.NN_t.requires_grad_(False)
                self.NN_logs.requires_grad_(False)
                nn.init.zeros_(self.CN.weight)
                nn.init.zeros_(self.CN.bias)
        self._init_done = False          # host mirror of the flag: no device sync per call

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._init_done = bool(int(self.initialized.item()))

    def is_initialized(self):
        return self._init_done

    # data-parallel first call: True = every rank contributes its shard's per-channel sums and all ranks initialise from
    # the GLOBAL batch statistics (contextflow_amd.dist.sharded_actnorm_init); False = each process uses what it sees
    sharded_init = False

    def initialize(self, x):
        """actnorm.py:28-35 on the device: two-stage fp64 reduction of sum x and sum x^2 per channel."""
        x, xbs = _hip.bview(x)
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        ws = torch.empty(_hip.lib().cf_actnorm_stats_ws_bytes(C), device=x.device, dtype=torch.uint8)
        t = torch.empty(C, device=x.device, dtype=torch.float32)
        logs = torch.empty(C, device=x.device, dtype=torch.float32)
        if ActNorm.sharded_init:
            from ..dist import allreduce_actnorm_sums
            sums = torch.zeros(2 * C + 1, device=x.device, dtype=torch.float64)     # [sum x | sum x^2 | elements per channel]
            if B > 0:
                _hip.call("cf_actnorm_sums", _hip.p(x), _hip.p(sums), _hip.p(ws), B, C, HW, xbs, _hip.stream())
                sums[2 * C] = float(B * HW)
            allreduce_actnorm_sums(sums)
            _hip.call("cf_actnorm_from_sums", _hip.p(sums), _hip.p(sums[2 * C:]), 0.0, _hip.p(t), _hip.p(logs), C, _hip.stream())
        else:
            _hip.call("cf_actnorm_stats", _hip.p(x), _hip.p(t), _hip.p(logs), _hip.p(ws), B, C, HW, xbs, _hip.stream())
        with torch.no_grad():
            self.NN_t.copy_(t)
            self.NN_logs.copy_(logs)
            self.initialized.fill_(1)
        self._init_done = True

    def _run(self, x, inverse):
        _hip.require_device(x, self.NN_t)
        x = _hip.f32(x)
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // max(B * C, 1) if B else 1
        out = torch.empty_like(x)
        s = None if inverse else torch.empty(1, device=x.device, dtype=torch.float32)
        _hip.call("cf_actnorm", _hip.p(x), _hip.p(_hip.f32(self.NN_t.detach())), _hip.p(_hip.f32(self.NN_logs.detach())),
                  _hip.p(out), _hip.p(s), B, C, HW, int(inverse), _hip.stream())
        return out, s

    def _forward_ctx(self, x, context, tape=None, pre=None):
        """actnorm.py:40-60: per-sample shift / log-scale CN(c), added to the shared ones under contextflow (the only
        branch that runs the data-dependent init).  pre: code, log-density and CN(c) from the grouped front end (specialist.py)."""
        from .simple_vit import _linear
        if pre is not None:
            c, logp_c = pre["c"], pre["logp"]
        else:
            c, logp_c = self.context_net(context)
        if self.contextflow and not self._init_done:
            self.initialize(x)
        x, xbs = _hip.bview(x)
        B, C, H, W = x.shape
        m = pre["m"] if pre is not None else _linear(_hip.f32(c), self.CN)      # (B, 2C)
        t = _hip.f32(self.NN_t.detach()) if self.contextflow else None
        logs = _hip.f32(self.NN_logs.detach()) if self.contextflow else None
        z = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
        ldj = torch.empty(B, device=x.device, dtype=torch.float32)
        _hip.call("cf_actnorm_ctx", _hip.p(x), _hip.p(m), _hip.p(t), _hip.p(logs), _hip.p(z), _hip.p(ldj), B, C, H * W, xbs,
                  _hip.stream())
        if tape is not None:
            tape.append(dict(x=x, c=_hip.f32(c), m=m, eps=encoder_noise(self.context_net)))
        return z, ldj + logp_c * float(H * W)

    def forward(self, x, context=None):
        _hip.require_device(x)
        if self.context_net:
            return self._forward_ctx(x, context)
        if not self._init_done:
            self.initialize(x)
        z, s = self._run(x, False)
        return z, s.expand(x.shape[0])

    def reverse(self, z, context=None):
        if self.context_net:
            raise NotImplementedError("ActNorm.reverse with a context net (the reference's own is marked 'to update')")
        assert self._init_done
        return self._run(z, True)[0]

    def logdet(self, x, context=None):
        return self.forward(x, context)[1]


class ActNormFC(ActNorm):
    def __init__(self, data_size):
        super().__init__(data_size)

    def forward(self, x, context=None):
        out, ldj = super().forward(x.view(-1, self.D, 1, 1), context)
        return out.view(-1, self.D), ldj

    def reverse(self, x, context=None):
        return super().reverse(x.view(-1, self.D, 1, 1), context).view(-1, self.D)

    def logdet(self, x, context=None):
        return super().logdet(x.view(-1, self.D, 1, 1))
