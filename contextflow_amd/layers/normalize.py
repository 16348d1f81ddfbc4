"""Normalization (reference: contextflow/layers/normalize.py:6-49): out = in/scale + translation."""
import math

import torch
from torch import Tensor

from . import _hip
from .flowlayer import PreprocessingFlowLayer


def _as_tensor(v):
    if isinstance(v, Tensor):
        return v
    if isinstance(v, (list, tuple)):
        return torch.Tensor(v)
    return torch.Tensor([v])


class Normalization(PreprocessingFlowLayer):
    def __init__(self, translation, scale, learnable=False):
        super().__init__()
        translation, scale = _as_tensor(translation), _as_tensor(scale)
        if learnable:
            self.translation = torch.nn.Parameter(translation)
            self.scale = torch.nn.Parameter(scale)
        else:
            self.register_buffer("translation", translation)
            self.register_buffer("scale", scale)
        if scale.numel() != 1 or translation.numel() != 1:
            raise NotImplementedError("contextflow_amd Normalization: only scalar scale/translation (model.py:98-99)")
        self._refresh()

    def _refresh(self):
        # host copies of the two scalars; refreshed whenever a state_dict is loaded
        self._t, self._s = float(self.translation.item()), float(self.scale.item())

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._refresh()

    def _run(self, input, inverse):
        _hip.require_device(input)
        x = _hip.f32(input)
        out = torch.empty_like(x)
        _hip.call("cf_affine", _hip.p(x), _hip.p(out), x.numel(), self._t, self._s, int(inverse), _hip.stream())
        return out

    def forward(self, input, context=None):
        return self._run(input, False), self.logdet(input, context)

    def reverse(self, input, context=None):
        return self._run(input, True)

    def logdet(self, input, context=None):
        # normalize.py:42-49 with a scalar scale: -C * (H*W) * log(scale), same for every sample
        B, C = input.shape[:2]
        n = input.numel() / B / C
        return input.new_full((B,), -C * n * math.log(self._s), dtype=torch.float32)
