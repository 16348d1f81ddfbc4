"""LogitTransform (reference: contextflow/layers/transforms.py:6-18)."""
import torch

from . import _hip
from .flowlayer import PreprocessingFlowLayer


class LogitTransform(PreprocessingFlowLayer):
    def forward(self, input, context=None):
        _hip.require_device(input)
        x = _hip.f32(input)
        B = x.shape[0]
        out = torch.empty_like(x)
        ldj = torch.empty(B, device=x.device, dtype=torch.float32)
        _hip.call("cf_logit_fwd", _hip.p(x), _hip.p(out), _hip.p(ldj), B, x.numel() // max(B, 1), _hip.stream())
        return out, ldj

    def reverse(self, input, context=None):
        _hip.require_device(input)
        x = _hip.f32(input)
        out = torch.empty_like(x)
        _hip.call("cf_sigmoid", _hip.p(x), _hip.p(out), x.numel(), _hip.stream())
        return out

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]
