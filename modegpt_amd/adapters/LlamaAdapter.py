"""Llama adapter (reference: src/adapters/LlamaAdapter.py).  Hook sites as the reference registers them
(LlamaAdapter.py:71-100): pre-hook on mlp.down_proj, forward hooks on input_layernorm, q_proj, k_proj
(pre-RoPE statistics; the post-RoPE variant is disabled upstream)."""
from __future__ import annotations

import copy
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from .model_adapter import (AttentionComponents, MLPComponents, MLPTensors, ModelAdapter, QKComponents, QKTensors,
                            VOComponents, VOTensors)


class LlamaAdapter(ModelAdapter):
    @property
    def arch(self) -> str:
        return "llama"

    def get_transformer_blocks(self) -> nn.ModuleList:
        return self.model.model.layers

    def _block(self, layer_idx: int) -> nn.Module:
        return self.get_transformer_blocks()[layer_idx]

    # ---- hooks ----
    def register_hooks(self, layer_idx, block, cov_mlp_list, cov_q_list, cov_k_list, cov_x_list, handles, logger):
        """Same four hook sites as the reference.  The three attention-side statistics are not launched when their hook
        fires: the hook only parks a reference to the activation, and the layer's last hook (the pre-hook of
        mlp.down_proj) sends all four problems of the layer to the device in ONE fused launch
        (ops.cov_accum_multi -> mdg_cov_accum_multi), so the small problems' workgroups run in the slots the large one
        leaves free.  Nothing is copied; the parked tensors live until the end of the layer's forward."""
        parked = {}

        def park(kind):
            @torch.no_grad()
            def hook(module, inp, out):
                parked[kind] = out
            return hook

        @torch.no_grad()
        def flush(module, input):
            items = [(cov_mlp_list[layer_idx], input[0], 1)]
            for kind, lst, heads in (("x", cov_x_list, 1), ("q", cov_q_list, self.n_heads), ("k", cov_k_list, self.n_kv_heads)):
                t = parked.pop(kind, None)
                if t is not None:
                    items.append((lst[layer_idx], t, heads))
            ops.cov_accum_multi(items)
            return None

        handles.append(block.input_layernorm.register_forward_hook(park("x")))
        handles.append(block.self_attn.k_proj.register_forward_hook(park("k")))
        handles.append(block.self_attn.q_proj.register_forward_hook(park("q")))
        handles.append(block.mlp.down_proj.register_forward_pre_hook(flush))

    # The reference's per-statistic hook factories, kept for adapters that register them one by one.
    @staticmethod
    def _llama_pre_gate_hook(layer_idx, cov_mlp_list):
        """sigma_mlp += H^T H, H = the input of down_proj (LlamaAdapter.py:127-136)."""
        @torch.no_grad()
        def hook(module, input: Tuple[Tensor]):
            ops.cov_accum(cov_mlp_list[layer_idx], input[0])
            return None
        return hook

    @staticmethod
    def _make_proj_hook(layer_idx, cov_list, n_heads, head_dim, d_model):
        """sigma_q / sigma_k += per-head Gram of a projection's pre-RoPE output (LlamaAdapter.py:115-125).  register_hooks
        above parks the projection outputs and flushes them in one fused launch instead; this stand-alone hook does the
        same accumulation for callers that register it themselves."""
        @torch.no_grad()
        def hook(module, inp, out):
            ops.cov_accum(cov_list[layer_idx], out, n_heads=n_heads)
        return hook

    @staticmethod
    def _input_hook(layer_idx, cov_list):
        """sigma_x += sum_b X_b^T X_b, X = input_layernorm's output (LlamaAdapter.py:138-147)."""
        @torch.no_grad()
        def hook(module, inp, out):
            ops.cov_accum(cov_list[layer_idx], out)
        return hook

    def compute_layer_energy(self, layer_idx: int, Ca: Optional[Tensor] = None) -> MLPTensors:
        raise NotImplementedError("compute_layer_energy not imp for llama")

    def calibrate_model(self, n_samples: int, batch_size: int, target_layers, dataset="wikitext"):
        raise NotImplementedError("custom calibrate model not impl for llama")

    # ---- components ----
    def get_mlp_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> MLPComponents:
        b = self._block(layer_idx)
        return MLPComponents(block=b, up_proj=b.mlp.up_proj, down_proj=b.mlp.down_proj, gate_proj=b.mlp.gate_proj)

    def get_mlp_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> MLPTensors:
        b = self._block(layer_idx)
        return MLPTensors(up_proj=b.mlp.up_proj.weight, down_proj=b.mlp.down_proj.weight,
                          gate_proj=b.mlp.gate_proj.weight)

    def get_vo_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> VOComponents:
        b = self._block(layer_idx)
        return VOComponents(block=b, v_proj=b.self_attn.v_proj, o_proj=b.self_attn.o_proj)

    def get_vo_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> VOTensors:
        b = self._block(layer_idx)
        return VOTensors(v_proj=b.self_attn.v_proj.weight, o_proj=b.self_attn.o_proj.weight)

    def get_qk_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> QKComponents:
        b = self._block(layer_idx)
        return QKComponents(block=b, query_proj=b.self_attn.q_proj, key_proj=b.self_attn.k_proj)

    def get_qk_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> QKTensors:
        b = self._block(layer_idx)
        return QKTensors(query_proj=b.self_attn.q_proj.weight, key_proj=b.self_attn.k_proj.weight)

    def get_attn_components(self, layer_idx: int) -> AttentionComponents:
        a = self._block(layer_idx).self_attn
        return AttentionComponents(block=self._block(layer_idx), q_proj=a.q_proj, k_proj=a.k_proj, v_proj=a.v_proj,
                                   o_proj=a.o_proj)

    def get_qk_weights(self, layer_idx: int) -> Tuple[Tensor, Tensor]:
        a = self._block(layer_idx).self_attn
        return a.q_proj.weight, a.k_proj.weight

    def get_vo_weights(self, layer_idx: int) -> Tuple[Tensor, Tensor]:
        a = self._block(layer_idx).self_attn
        return a.v_proj.weight, a.o_proj.weight

    # ---- rebuild ----
    def replace_mlp_layers(self, layer_idx, new_up, new_down, new_gate=None, expert_idx=None) -> None:
        mlp = self._block(layer_idx).mlp
        mlp.up_proj, mlp.down_proj = new_up, new_down
        if new_gate is not None:
            mlp.gate_proj = new_gate

    def replace_attn_layers(self, layer_idx, new_q, new_k, new_v, new_o) -> None:
        a = self._block(layer_idx).self_attn
        for name, mod in (("q_proj", new_q), ("k_proj", new_k), ("v_proj", new_v), ("o_proj", new_o)):
            if mod is not None:
                setattr(a, name, mod)

    def patch_config(self):
        """Stamp the per-layer ranks and the auto_map the patched modeling files expect (LlamaAdapter.py:250-302)."""
        cfg = self.model.config
        original = copy.deepcopy(cfg)
        ranks = {k: [] for k in ("q", "k", "v", "o", "gate")}
        for i in range(cfg.num_hidden_layers):
            qk, vo, mlp = self.get_qk_tensors(i), self.get_vo_tensors(i), self.get_mlp_tensors(i)
            ranks["q"].append(qk.query_proj.shape[0])
            ranks["k"].append(qk.key_proj.shape[0])
            ranks["v"].append(vo.v_proj.shape[0])
            ranks["o"].append(vo.o_proj.shape[1])
            ranks["gate"].append(mlp.gate_proj.shape[0])
        cfg.ffn_dim = -1
        cfg.q_ranks, cfg.k_ranks, cfg.v_ranks, cfg.o_ranks = ranks["q"], ranks["k"], ranks["v"], ranks["o"]
        cfg.gate_ranks = ranks["gate"]
        mt = cfg.model_type
        if mt == "opt":
            cfg.auto_map = {"AutoModelForCausalLM": "OPTRebuild.OPTForCausalLM"}
        if mt == "llama":
            cfg.auto_map = {"AutoModelForCausalLM": "LlamaRebuild.LlamaForCausalLM"}
        if "qwen" in mt:
            cfg.auto_map = {"AutoModelForCausalLM": "DenseQwenRebuild.Qwen3ForCausalLM"}
        return original
