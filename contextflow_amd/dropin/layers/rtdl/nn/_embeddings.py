"""The names contextflow/model.py:14 star-imports from rtdl/nn/_embeddings.py and uses (model.py:34,39,46)."""
from contextflow_amd.layers.context import CatEmbeddings, EyeEncoder, OneHotEncoder

__all__ = ["CatEmbeddings", "EyeEncoder", "OneHotEncoder"]
