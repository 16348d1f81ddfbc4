"""Drop-in for the reference's `layers` package (contextflow/layers/__init__.py:1-14) on MI355X.

Same class names, constructor signatures, `forward/reverse/logdet` contract and `state_dict` keys;
underneath, every arithmetic op of the density path is a hand-written gfx950 kernel reached through
the C ABI in include/contextflow_hip.h.

To resolve sub-packages this package does not re-implement (the reference's vendored `layers.rtdl`),
set CONTEXTFLOW_REFERENCE_LAYERS=/path/to/contextflow/layers: it is appended to this package's
search path."""
import os as _os

from .flowlayer import FlowLayer, PreprocessingFlowLayer, ModifiedGradFlowLayer
from .dequantize import Dequantization
from .normalize import Normalization
from .augment import Augment
from .distributions import StandardNormal, GaussianDistribution, GaussianMixtureDistribution, UniformDistribution
from .splitprior import SplitPrior
from .flowsequential import FlowSequential, FlowInvSequential, GraphedFlow
from .conv1x1 import Conv1x1, FC
from .activations import (FlowActivationLayer, Identity, LeakyRelu, LearnableLeakyRelu, Sigmoid, SmoothLeakyRelu, SmoothTanh,
                          SplineActivation)
from .actnorm import ActNorm, ActNormFC
from .squeeze import Squeeze, UnSqueeze
from .transforms import LogitTransform
from .coupling import Coupling, CouplingFC, TransCoupling
from .simple_vit import SimpleViT, posemb_sincos_2d
from .ar import MaskedCoupling
from .permute_axes import PermuteAxes
from .context import (ArgmaxCatDequantization, CatEmbeddings, ConditionalGaussianDistribution, ContextEncoder,
                      EyeEncoder, EyeSampling, OneHotEncoder, ProbSampling, UniformCatDequantization,
                      VariationalCatDequantization)
from .unsupported import *  # noqa: F401,F403

_ref = _os.environ.get("CONTEXTFLOW_REFERENCE_LAYERS")
if _ref and _os.path.isdir(_ref):
    __path__.append(_ref)
