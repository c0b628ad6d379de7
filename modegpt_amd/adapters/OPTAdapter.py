"""OPT adapter with the reference's INTENDED semantics restored (src/adapters/OPTAdapter.py is bit-rotted upstream,
SURVEY.md 9-O): sigma_mlp from ReLU(fc1 out), per-head q/k statistics, sigma_x from the pre-attention layer norm,
no gate matrix, MHA VO path, OPT CR score."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from .model_adapter import (AttentionComponents, MLPComponents, MLPTensors, ModelAdapter, QKComponents, QKTensors,
                            VOComponents, VOTensors)


class OPTAdapter(ModelAdapter):
    @property
    def arch(self) -> str:
        return "opt"

    def get_transformer_blocks(self) -> nn.ModuleList:
        return self.model.model.decoder.layers

    def _block(self, i):
        return self.get_transformer_blocks()[i]

    def get_n_inner(self, block=None) -> int:
        b = block if block is not None else self._block(0)
        return b.fc1.out_features

    @property
    def d_int(self) -> int:
        return self.get_n_inner()

    def register_hooks(self, layer_idx, block, cov_mlp_list, cov_q_list, cov_k_list, cov_x_list, handles, logger):
        handles.append(block.fc1.register_forward_hook(self._make_fc_hook(layer_idx, cov_mlp_list)))
        handles.append(block.self_attn.q_proj.register_forward_hook(
            self._make_proj_hook(layer_idx, cov_q_list, self.n_heads, self.head_dim, self.d_model)))
        handles.append(block.self_attn.k_proj.register_forward_hook(
            self._make_proj_hook(layer_idx, cov_k_list, self.n_heads, self.head_dim, self.d_model)))
        # the reference's on_batch_end_step (OPTAdapter.py:54-55) is never called; the statistic it meant is the
        # input of the attention block, i.e. self_attn_layer_norm's output
        handles.append(block.self_attn_layer_norm.register_forward_hook(self._x_hook(layer_idx, cov_x_list)))

    @staticmethod
    def _x_hook(layer_idx, cov_x_list):
        @torch.no_grad()
        def hook(module, inp, out):
            ops.cov_accum(cov_x_list[layer_idx], out)
        return hook

    def on_batch_end_step(self, layer_idx, x_in, cov_x_list):
        ops.cov_accum(cov_x_list[layer_idx], x_in)

    def compute_layer_energy(self, layer_idx: int, Ca: Optional[Tensor] = None) -> MLPTensors:
        raise NotImplementedError("compute_layer_energy not imp for opt")

    def calibrate_model(self, n_samples: int, batch_size: int, target_layers, dataset="wikitext"):
        raise NotImplementedError("custom calibrate model not impl for opt")

    def get_mlp_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> MLPComponents:
        b = self._block(layer_idx)
        return MLPComponents(block=b, up_proj=b.fc1, down_proj=b.fc2, gate_proj=None)

    def get_mlp_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> MLPTensors:
        b = self._block(layer_idx)
        return MLPTensors(up_proj=b.fc1.weight, down_proj=b.fc2.weight, gate_proj=None)

    def get_vo_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> VOComponents:
        b = self._block(layer_idx)
        return VOComponents(block=b, v_proj=b.self_attn.v_proj, o_proj=b.self_attn.out_proj)

    def get_vo_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> VOTensors:
        b = self._block(layer_idx)
        return VOTensors(v_proj=b.self_attn.v_proj.weight, o_proj=b.self_attn.out_proj.weight)

    def get_qk_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> QKComponents:
        b = self._block(layer_idx)
        return QKComponents(block=b, query_proj=b.self_attn.q_proj, key_proj=b.self_attn.k_proj)

    def get_qk_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> QKTensors:
        b = self._block(layer_idx)
        return QKTensors(query_proj=b.self_attn.q_proj.weight, key_proj=b.self_attn.k_proj.weight)

    def get_attn_components(self, layer_idx: int) -> AttentionComponents:
        a = self._block(layer_idx).self_attn
        return AttentionComponents(block=self._block(layer_idx), q_proj=a.q_proj, k_proj=a.k_proj, v_proj=a.v_proj,
                                   o_proj=a.out_proj)

    def get_qk_weights(self, layer_idx: int) -> Tuple[Tensor, Tensor]:
        a = self._block(layer_idx).self_attn
        return a.q_proj.weight, a.k_proj.weight

    def get_vo_weights(self, layer_idx: int) -> Tuple[Tensor, Tensor]:
        a = self._block(layer_idx).self_attn
        return a.v_proj.weight, a.out_proj.weight

    def replace_mlp_layers(self, layer_idx, new_up, new_down, new_gate=None, expert_idx=None) -> None:
        b = self._block(layer_idx)
        b.fc1, b.fc2 = new_up, new_down

    def replace_attn_layers(self, layer_idx, new_q, new_k, new_v, new_o) -> None:
        a = self._block(layer_idx).self_attn
        for name, mod in (("q_proj", new_q), ("k_proj", new_k), ("v_proj", new_v), ("out_proj", new_o)):
            if mod is not None:
                setattr(a, name, mod)

    def patch_config(self):
        """Ranks the OPT patcher reads (OPTRebuild.py:126-127,241-242 expect qk_ranks / vo_ranks)."""
        import copy
        cfg = self.model.config
        original = copy.deepcopy(cfg)
        cfg.qk_ranks = [self.get_qk_tensors(i).query_proj.shape[0] for i in range(self.n_layers)]
        cfg.vo_ranks = [self.get_vo_tensors(i).v_proj.shape[0] for i in range(self.n_layers)]
        cfg.gate_ranks = [self.get_mlp_tensors(i).up_proj.shape[0] for i in range(self.n_layers)]
        cfg.ffn_dim = -1
        cfg.auto_map = {"AutoModelForCausalLM": "OPTRebuild.OPTForCausalLM"}
        return original
