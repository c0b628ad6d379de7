"""Modeling file copied next to a compressed OPT checkpoint; config.auto_map names `OPTRebuild.OPTForCausalLM`.
Stock HF classes with q/k/v/out_proj and fc1/fc2 resized to config.qk_ranks / vo_ranks / gate_ranks (bias-free, as
ModelAdapter.convert_model builds them); attention views q/k and v with their own head widths and pre-scales q by
the compressed width (reference semantics: src/patchers/OPTRebuild.py:120-163)."""
from transformers.models.opt.modeling_opt import OPTForCausalLM as _StockOPTForCausalLM

from .compressed_attention import shrink_to_config_ranks   # travels with the checkpoint (save_compressed_model copies both files)


class OPTForCausalLM(_StockOPTForCausalLM):
    def __init__(self, config):
        poisoned = config.ffn_dim           # patch_config stores -1 on purpose (src/patchers/patch.py:57-58)
        if poisoned is None or poisoned < 0:
            config.ffn_dim = max(config.gate_ranks)   # any valid width: every fc1 / fc2 is resized just below
        super().__init__(config)
        config.ffn_dim = poisoned
        shrink_to_config_ranks(self, "opt")
