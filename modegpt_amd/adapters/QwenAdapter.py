"""Dense Qwen3 = the Llama adapter under arch "qwen3" (reference: src/adapters/QwenAdapter.py:6-9); statistics
are taken before q_norm / k_norm, as the q_proj / k_proj output hooks see them."""
from .LlamaAdapter import LlamaAdapter


class QwenAdapter(LlamaAdapter):
    @property
    def arch(self) -> str:
        return "qwen3"
