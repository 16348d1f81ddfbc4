"""Dequantization (reference: contextflow/layers/dequantize.py:8-23)."""
import torch

from . import _hip
from .flowlayer import PreprocessingFlowLayer


class Dequantization(PreprocessingFlowLayer):
    def __init__(self, dist):
        super().__init__()
        self.dist = dist          # a distribution with support on [0, 1]^d

    def forward(self, input, context=None):
        _hip.require_device(input)
        noise, log_qnoise = self.dist.sample(input.size(0), context=input)
        x, u = _hip.f32(input), _hip.f32(noise)
        out = torch.empty_like(x)
        _hip.call("cf_dequant_fwd", _hip.p(x), _hip.p(u), _hip.p(out), x.numel(), _hip.stream())
        return out, -log_qnoise

    def reverse(self, input, context=None):
        _hip.require_device(input)
        x = _hip.f32(input)
        out = torch.empty_like(x)
        _hip.call("cf_floor", _hip.p(x), _hip.p(out), x.numel(), _hip.stream())
        return out

    def logdet(self, input, context=None):
        raise NotImplementedError
