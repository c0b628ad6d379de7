"""Modeling file copied next to a compressed Llama checkpoint; config.auto_map names `LlamaRebuild.LlamaForCausalLM`
(what the reference's patch_config writes, src/patchers/patch.py:66-67 / LlamaAdapter.py:286-300), so
`AutoModelForCausalLM.from_pretrained(dir, trust_remote_code=True)` lands here.

Unlike the reference's file of the same name (a fork of the whole HF Llama implementation), this one keeps the stock HF
classes: the constructor resizes every projection to the per-layer ranks recorded in the config, loads the rotary masks
from config.mask_path and installs the compressed attention forward, whose elementwise chain is one HIP kernel
(modegpt_amd/patchers/compressed_attention.py, csrc/rope.hip).  The state-dict keys are the stock ones, so a checkpoint
written by either engine loads.
"""
from transformers.models.llama.modeling_llama import LlamaForCausalLM as _StockLlamaForCausalLM

from .compressed_attention import shrink_to_config_ranks   # travels with the checkpoint (save_compressed_model copies both files)


class LlamaForCausalLM(_StockLlamaForCausalLM):
    def __init__(self, config):
        super().__init__(config)
        shrink_to_config_ranks(self, "llama")
