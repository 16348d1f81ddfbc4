from .gaussian import StandardNormal, GaussianDistribution, GaussianMixtureDistribution, gmm_prepare, gmm_logprob
from .uniform import UniformDistribution

__all__ = ["StandardNormal", "GaussianDistribution", "GaussianMixtureDistribution", "UniformDistribution"]
