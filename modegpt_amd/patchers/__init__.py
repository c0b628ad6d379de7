from .compressed_attention import install_compressed_attention  # noqa: F401
