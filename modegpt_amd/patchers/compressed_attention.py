"""In-process attention for a model whose projections were swapped by `convert_model`.

A compressed layer keeps r_qk <= head_dim columns of every q/k head (chosen per kv head, in RoPE pairs, order given by
the layer's rotary mask) and r_vo columns of every v/o head.  The stock HF attention modules assume `head_dim`
everywhere, so the reference ships forked modeling files with the checkpoint (src/patchers/*Rebuild.py, loaded through
config.auto_map).  Those files are the reference's own and are not reproduced here; this module gives the SAME
semantics to the live model object so the compressed model can be evaluated right after `convert_model`:

  * q/k/v are viewed with their own per-layer head widths                        (LlamaRebuild.py:320-326)
  * RoPE: cos/sin are gathered along the feature axis by the rotary mask, per kv head, query heads of a group share
    their kv head's mask; rotate_half then pairs the two halves of the KEPT columns  (LlamaRebuild.py:153-176)
  * Qwen3: q_norm / k_norm normalise over the kept columns with the norm weight gathered by the same mask
                                                                                 (DenseQwenRebuild.py:262-286)
  * softmax scale = (compressed q/k head width) ** -0.5                          (LlamaRebuild.py:266,282)
  * OPT: no RoPE, no mask; q is pre-scaled by the compressed width               (OPTRebuild.py:144-146)
"""
from __future__ import annotations

import types
from typing import List, Optional

import torch
import torch.nn.functional as F


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _sdpa(q, k, v, attention_mask, scale: float, n_rep: int):
    if n_rep > 1:
        k = k.repeat_interleave(n_rep, dim=1)
        v = v.repeat_interleave(n_rep, dim=1)
    mask = attention_mask if (torch.is_tensor(attention_mask) and attention_mask.dim() == 4) else None
    if mask is not None:
        mask = mask[..., : k.shape[-2]]
    causal = mask is None and q.shape[-2] > 1 and q.shape[-2] == k.shape[-2]
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=causal, scale=scale)


def _rope_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
    """Llama / Qwen3 (RoPE, optional GQA, optional per-head q/k RMSNorm)."""
    B, T, _ = hidden_states.shape
    n_h, n_kv = self.config.num_attention_heads, self.config.num_key_value_heads
    q = self.q_proj(hidden_states)
    k = self.k_proj(hidden_states)
    v = self.v_proj(hidden_states)
    r_qk, r_vo = q.shape[-1] // n_h, v.shape[-1] // n_kv
    q = q.view(B, T, n_h, r_qk)
    k = k.view(B, T, n_kv, r_qk)
    v = v.view(B, T, n_kv, r_vo).transpose(1, 2)
    mask_k = self.layer_rotary_mask                              # [n_kv, r_qk] indices into the original head_dim
    if mask_k is not None and mask_k.device != q.device:
        mask_k = self.layer_rotary_mask = mask_k.to(q.device)
    mask_q = None if mask_k is None else mask_k.repeat_interleave(n_h // n_kv, dim=0)   # [n_h, r_qk]
    if getattr(self, "q_norm", None) is not None:                # Qwen3: RMSNorm over the kept columns, gathered weight
        def masked_norm(x, norm, m):
            xf = x.float()
            xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + norm.variance_epsilon)
            w = norm.weight if m is None else norm.weight[m]     # [heads, r_qk]
            return (w * xf).to(x.dtype)
        q = masked_norm(q, self.q_norm, mask_q)
        k = masked_norm(k, self.k_norm, mask_k)
    q = q.transpose(1, 2)                                        # [B, n_h, T, r_qk]
    k = k.transpose(1, 2)
    cos, sin = position_embeddings                               # [B, T, head_dim]
    if mask_k is None:
        cq = ck = cos.unsqueeze(1)
        sq = sk = sin.unsqueeze(1)
    else:
        ck, sk = cos[:, :, mask_k].permute(0, 2, 1, 3), sin[:, :, mask_k].permute(0, 2, 1, 3)   # [B, n_kv, T, r_qk]
        cq, sq = cos[:, :, mask_q].permute(0, 2, 1, 3), sin[:, :, mask_q].permute(0, 2, 1, 3)   # [B, n_h,  T, r_qk]
    q = q * cq + _rotate_half(q) * sq
    k = k * ck + _rotate_half(k) * sk
    if past_key_values is not None:
        k, v = past_key_values.update(k, v, self.layer_idx)
    out = _sdpa(q, k, v, attention_mask, float(r_qk) ** -0.5, n_h // n_kv)
    out = out.transpose(1, 2).reshape(B, T, n_h * r_vo)
    return self.o_proj(out), None


def _opt_forward(self, hidden_states, past_key_values=None, attention_mask=None, output_attentions=False, **kwargs):
    B, T, _ = hidden_states.shape
    n_h = self.num_heads
    q = self.q_proj(hidden_states)
    k = self.k_proj(hidden_states)
    v = self.v_proj(hidden_states)
    r_qk, r_vo = q.shape[-1] // n_h, v.shape[-1] // n_h
    q = q.view(B, T, n_h, r_qk).transpose(1, 2)
    k = k.view(B, T, n_h, r_qk).transpose(1, 2)
    v = v.view(B, T, n_h, r_vo).transpose(1, 2)
    if past_key_values is not None:
        k, v = past_key_values.update(k, v, self.layer_idx)
    out = _sdpa(q, k, v, attention_mask, float(r_qk) ** -0.5, 1)
    out = out.transpose(1, 2).reshape(B, T, n_h * r_vo)
    return self.out_proj(out), None


def install_compressed_attention(adapter, rotary_masks: Optional[List[torch.Tensor]]) -> None:
    """Give every attention module of `adapter.model` the compressed-aware forward.  `rotary_masks`: the list
    `compress_qk` returned (one int64 [n_kv, r_qk] tensor per layer, layer order) or None for architectures without
    RoPE masks (OPT) / for an uncompressed QK stage."""
    blocks = adapter.get_transformer_blocks()
    for i, block in enumerate(blocks):
        attn = block.self_attn
        if adapter.arch == "opt":
            attn.forward = types.MethodType(_opt_forward, attn)
        else:
            attn.layer_rotary_mask = None if rotary_masks is None else rotary_masks[i]
            attn.forward = types.MethodType(_rope_forward, attn)
