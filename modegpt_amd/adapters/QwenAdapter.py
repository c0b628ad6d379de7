"""Dense Qwen3.

Structurally a Llama block (q/k/v/o projections, gate/up/down MLP, RMSNorm before attention), so every hook site,
component getter and rebuild method of LlamaAdapter applies unchanged; only the architecture tag differs, which is
what `compress_qk` / `compress_vo` / `patch_config` / `save_compressed_model` key on ("qwen" in arch selects the
RoPE-pair CR scoring, the even-rank rule and the DenseQwenRebuild modeling file).  Qwen3 applies q_norm / k_norm
AFTER the projections; the statistics are taken from the raw q_proj / k_proj outputs, i.e. before those norms --
the same choice the reference makes (src/adapters/QwenAdapter.py:6-9 on top of LlamaAdapter.py:91-100).
"""
from .LlamaAdapter import LlamaAdapter


class QwenAdapter(LlamaAdapter):
    ARCH_TAG = "qwen3"

    @property
    def arch(self) -> str:
        return self.ARCH_TAG
