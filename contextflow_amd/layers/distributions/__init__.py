from .gaussian import StandardNormal, GaussianMixtureDistribution, gmm_prepare, gmm_logprob
from .uniform import UniformDistribution

__all__ = ["StandardNormal", "GaussianMixtureDistribution", "UniformDistribution"]
