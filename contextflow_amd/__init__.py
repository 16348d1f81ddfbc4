"""contextflow_amd — MI355X-native (gfx950) implementation of ContextFlow's coupling-layer
density-estimation path behind the reference's `layers.*` nn.Module API.

    from contextflow_amd import layers, create_model
"""
from . import layers
from . import optim
from .model import create_model, preset_config, PRESETS

__all__ = ["layers", "optim", "create_model", "preset_config", "PRESETS"]
