"""modegpt_amd -- MI355X-native engine for MoDeGPT's per-layer compression path.

Host side (this package) mirrors the reference's Python surface for the path: `load_calibs`,
`allocate_global_sparsity`, `compress_nystrom`, `compress_qk`, `compress_vo`, `sqrt_M`, `CompressionConfig`,
the `ModelAdapter` plug-in ABC and its Llama / Qwen3 / OPT adapters, and the `run_modegpt` driver.  All tensor
math is done by hand-written gfx950 kernels in `libmodegpt_hip.so` (see include/modegpt_hip.h) reached through
ctypes; there is no CPU fallback.
"""
__version__ = "0.1.0"
