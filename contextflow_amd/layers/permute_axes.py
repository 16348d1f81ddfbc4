"""PermuteAxes (reference: contextflow/layers/permute_axes.py:5-22): a fixed permutation of the non-batch axes - the
ATM topology swaps the channel and time axes after every flow step (model.py:149-151).  Pure index map, ldj = 0."""
import torch

from .flowlayer import FlowLayer


class PermuteAxes(FlowLayer):
    def __init__(self, permutation):
        super().__init__()
        permutation = tuple(permutation)
        assert permutation[0] == 0, "First element of permutation must be 0 (such that batch dimension stays intact)"
        self.permutation = permutation
        self.inverse_permutation = torch.argsort(torch.tensor(permutation)).tolist()

    def forward(self, input, context=None):
        return input.permute(self.permutation).contiguous(), self.logdet(input, context)

    def reverse(self, input, context=None):
        return input.permute(self.inverse_permutation).contiguous()

    def logdet(self, input, context=None):
        return input.new_zeros(len(input))
