from ._embeddings import CatEmbeddings, EyeEncoder, OneHotEncoder  # noqa: F401
