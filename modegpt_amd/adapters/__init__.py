from .CompressionConfig import CompressionConfig  # noqa: F401
