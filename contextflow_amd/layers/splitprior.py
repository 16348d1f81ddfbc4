"""SplitPrior (reference: contextflow/layers/splitprior.py:7-25): second channel half is scored by a
prior and leaves the flow; its log-density is returned as the layer's ldj (B, M)."""
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
        raise NotImplementedError("SplitPrior.reverse is broken in the reference (splitprior.py:18); "
                                  "sampling is a later scope row (SURVEY.md §8(f) rank 3)")

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]
