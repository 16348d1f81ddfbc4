"""Squeeze / UnSqueeze (reference: contextflow/layers/squeeze.py:5-32), space-to-depth index map."""
import torch

from . import _hip
from .flowlayer import FlowLayer


def squeeze_op(x, p, inverse):
    """inverse=False: (B,C,H,W) -> (B,C*p1*p2,H/p1,W/p2); inverse=True the opposite."""
    _hip.require_device(x)
    x, xbs = _hip.bview(x)
    B = x.shape[0]
    if not inverse:
        C, H, W = x.shape[1:]
        out = torch.empty(B, C * p[0] * p[1], H // p[0], W // p[1], device=x.device, dtype=torch.float32)
    else:
        C, H, W = x.shape[1] // (p[0] * p[1]), x.shape[2] * p[0], x.shape[3] * p[1]
        out = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
    _hip.call("cf_squeeze", _hip.p(x), _hip.p(out), B, C, H, W, p[0], p[1], xbs, C * H * W, int(inverse), _hip.stream())
    return out


class Squeeze(FlowLayer):
    def __init__(self, patch_size=(2, 2)):
        super().__init__()
        self.p = patch_size

    def forward(self, input, context=None):
        return squeeze_op(input, self.p, False), self.logdet(input, context)

    def reverse(self, input, context=None):
        return squeeze_op(input, self.p, True)

    def logdet(self, input, context=None):
        return input.new_zeros(len(input))


class UnSqueeze(FlowLayer):
    def __init__(self, patch_size=(2, 2)):
        super().__init__()
        self.p = patch_size

    def forward(self, input, context=None):
        return squeeze_op(input, self.p, True), self.logdet(input, context)

    def reverse(self, input, context=None):
        return squeeze_op(input, self.p, False)

    def logdet(self, input, context=None):
        return input.new_zeros(len(input))
