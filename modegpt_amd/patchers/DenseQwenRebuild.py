"""Modeling file copied next to a compressed Qwen3 (dense) checkpoint; config.auto_map names
`DenseQwenRebuild.Qwen3ForCausalLM`.  Same construction as LlamaRebuild.py: stock HF classes, projections resized to
the config's per-layer ranks, rotary masks from config.mask_path; q_norm / k_norm keep their full head_dim weight (so
the checkpoint loads) and are applied over the kept columns with the weight gathered by the rotary mask, inside the
same HIP kernel as the rotation (reference semantics: src/patchers/DenseQwenRebuild.py:246-286)."""
from transformers.models.qwen3.modeling_qwen3 import Qwen3ForCausalLM as _StockQwen3ForCausalLM

from .compressed_attention import shrink_to_config_ranks   # travels with the checkpoint (save_compressed_model copies both files)


class Qwen3ForCausalLM(_StockQwen3ForCausalLM):
    def __init__(self, config):
        super().__init__(config)
        shrink_to_config_ranks(self, "qwen3")
