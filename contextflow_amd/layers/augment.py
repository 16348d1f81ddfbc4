"""Augment (reference: contextflow/layers/augment.py:6-26): pad the channel dim with noise."""
import torch

from .flowlayer import FlowLayer


class Augment(FlowLayer):
    def __init__(self, aug_distribution, aug_size, split_dim=1):
        super().__init__()
        self.distribution = aug_distribution
        self.aug_size = aug_size
        self.split_dim = split_dim

    def forward(self, input, context=None):
        noise, log_qnoise = self.distribution.sample(input.size(0))
        return torch.cat([input, noise], dim=self.split_dim), -log_qnoise

    def reverse(self, input, context=None):
        keep = input.shape[self.split_dim] - self.aug_size
        return input.narrow(self.split_dim, 0, keep)

    def logdet(self, input, context=None):
        raise NotImplementedError
