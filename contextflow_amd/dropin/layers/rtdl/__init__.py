"""Stand-in for the reference's vendored `layers.rtdl` (Yandex rtdl, contextflow/layers/rtdl/): only the context
encoders `create_model` uses are provided, from contextflow_amd.layers.context."""
from . import nn  # noqa: F401
