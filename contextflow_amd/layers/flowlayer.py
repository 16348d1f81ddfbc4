"""FlowLayer contract of the reference (contextflow/layers/flowlayer.py:7-51):
forward(input, context=None) -> (output, ldj); reverse(input, context=None) -> input; logdet(...) -> ldj.
`ldj` is (B,) or (B, M)."""
from abc import ABCMeta, abstractmethod

import torch.nn as nn


class FlowLayer(nn.Module, metaclass=ABCMeta):
    @abstractmethod
    def forward(self, input, context=None):
        ...

    @abstractmethod
    def reverse(self, input, context=None):
        ...

    @abstractmethod
    def logdet(self, input, context=None):
        ...


class PreprocessingFlowLayer(FlowLayer):
    pass


class ModifiedGradFlowLayer(FlowLayer):
    pass


def no_context(module_name, context_net):
    """The generalist (context-free) path is implemented natively; specialist variants are the next
    scope row."""
    if context_net:
        raise NotImplementedError(
            "%s: context-conditioned (specialist) variant is not implemented in contextflow_amd yet" % module_name)


def encoder_noise(context_net):
    """The Gaussian draw a flow-type encoder (vardeq / argmax / probsample) made in the call that just returned, None
    for the parameter-free encoders.  The training forward stores it in ITS tape record, so that a second forward before
    the backward (summed micro-batches, an evaluation pass in between) cannot change the noise the backward replays."""
    flow = getattr(context_net[1], "encoder", None)
    return getattr(getattr(flow, "dist", None), "last_eps", None)
