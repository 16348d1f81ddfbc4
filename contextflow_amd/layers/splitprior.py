"""SplitPrior (reference: contextflow/layers/splitprior.py:7-25): second channel half is scored by a
prior and leaves the flow; its log-density is returned as the layer's ldj (B, M)."""
import torch

from .flowlayer import FlowLayer


class SplitPrior(FlowLayer):
    def __init__(self, dist):
        super().__init__()
        self.dist = dist

    def forward(self, x, context=None):
        c = x.shape[1] // 2
        # channel slices are passed to the kernels with their batch stride: no copy
        return x[:, :c], self.dist.log_prob(x[:, c:], context)

    def reverse(self, z, context=None):
        """Inverse by specification: the reference's own line (splitprior.py:18, `self.dist.sample(self.C, ...)`)
        reads an attribute that is never set; the evident intent — resample the split-off half from its prior, one
        draw per batch element, and concatenate — is what runs here."""
        z2 = (self.dist.sample(z.shape[0], context, need_log_prob=False) if hasattr(self.dist, "mG") else self.dist.sample(z.shape[0], context))[0]
        return torch.cat([z, z2], dim=1)

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]
