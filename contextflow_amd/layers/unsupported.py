"""Names the reference's `from layers import *` exposes that are outside the density hot path
(§8(f)).  They exist so that model-assembly code importing them keeps
working; constructing one raises."""
import torch.nn as nn


def _stub(name, why):
    def __init__(self, *a, **k):
        raise NotImplementedError("%s is not implemented in contextflow_amd (%s)" % (name, why))
    return type(name, (nn.Module,), {"__init__": __init__})


StudentMixtureDistribution = _stub("StudentMixtureDistribution", "--dist tdist; non-default prior")
MultivariateGaussianMixtureDistribution = _stub("MultivariateGaussianMixtureDistribution", "unused by create_model")
