"""QK compression: CR column selection in RoPE-pair units (reference: src/compression/compress_qk.py:152-476)."""
from __future__ import annotations

import logging
from typing import List, Optional

import torch
from torch import Tensor

from .. import _lib, ops
from ..adapters.model_adapter import ModelAdapter
from ..model_utils import dtype_p, local_device

logger = logging.getLogger("MoDeGPT")

_SQRT_M_DEFAULT_RIDGE = 1e-4  # sqrt_M's default ridge_lambda (compression_utils.py:16)


def qk_rank_rule(head_dim: int, keep_ratio: float, arch: str, rank: Optional[int] = None) -> int:
    """compress_qk.py:176-182."""
    r = int(head_dim * keep_ratio) if rank is None else rank
    r = max(1, min(r, head_dim))
    if arch == "llama" or "qwen" in arch:
        r = r - (r % 2)
        r = max(2, min(r, head_dim))
    return r


def qk_mode_and_ridges(arch: str, grouped: bool, ridge_qk: float):
    """Which scoring variant compress_layer dispatches to, with the ridges each one really uses:
    GQA: K gets config.ridge_qk, Q gets sqrt_M's default; MHA llama / OPT: defaults for both."""
    if (arch == "llama" or "qwen" in arch) and grouped:
        return _lib.MDG_QK_ROPE_GROUPED, _SQRT_M_DEFAULT_RIDGE, ridge_qk
    if arch == "llama":
        return _lib.MDG_QK_ROPE_MHA, _SQRT_M_DEFAULT_RIDGE, _SQRT_M_DEFAULT_RIDGE
    if arch == "opt":
        return _lib.MDG_QK_OPT, _SQRT_M_DEFAULT_RIDGE, _SQRT_M_DEFAULT_RIDGE
    raise NotImplementedError("Most likely have to implement it compression for this model.")


@torch.no_grad()
def compress_layer(adapter: ModelAdapter, layer_idx: int, rank: int, cov_q_list: Tensor, cov_k_list: Tensor,
                   rotary_masks: List[Tensor], slice_dims=True, bias=True):
    """One layer (compress_qk.py:208-308): score, select (score-descending order, NOT sorted), gather the rows
    of W_q (all heads of the group) and W_k, append the [n_kv, rank] int64 rotary mask, save {"q_proj","k_proj"}."""
    n_heads, head_dim, arch, n_kv = adapter.n_heads, adapter.head_dim, adapter.arch, adapter.n_kv_heads
    comps = adapter.get_qk_components(layer_idx=layer_idx)
    mode, ridge_q, ridge_k = qk_mode_and_ridges(arch, n_kv != n_heads, adapter.config.ridge_qk)
    cq = cov_q_list.to(device=local_device(), dtype=dtype_p)
    ck = cov_k_list.to(device=local_device(), dtype=dtype_p)
    mask, q_rows, k_rows = ops.qk_select(cq, ck, rank, mode, ridge_q, ridge_k)
    W_q = comps.query_proj.weight.detach().to(device=local_device(), dtype=torch.bfloat16)
    W_k = comps.key_proj.weight.detach().to(device=local_device(), dtype=torch.bfloat16)
    Q_heads = ops.gather_rows(W_q, q_rows)   # [n_heads*rank, d]
    K_heads = ops.gather_rows(W_k, k_rows)   # [n_kv*rank, d]
    if (arch == "llama" or "qwen" in arch) and slice_dims:
        rotary_masks.append(mask)
    adapter.save_layer(output_dir=adapter.config.temp_storage_dir, suffix="qk",
                       weights={"q_proj": Q_heads, "k_proj": K_heads}, layer_idx=layer_idx)
    return mask


def _kept_rows(head: Tensor, rows: Tensor, slice_dims: bool) -> Tensor:
    """head[rows] when slicing; otherwise a zero matrix of the head's shape with those rows copied (the reference's
    slice_dims=False convention: same shape, dropped rows zeroed)."""
    if slice_dims:
        return head[rows]
    out = torch.zeros_like(head)
    out[rows] = head[rows]
    return out


@torch.no_grad()
def compress_head_llama_grouped(kv_head_idx: int, kv_head_ratio: int, cov_q_layer, cov_k_layer, Wq_heads: Tensor,
                                Wk_heads: Tensor, Q_heads_out: list, K_heads_out: list, layer_rotary_mask: list, rank: int,
                                slice_dims=True, ridge_lambda=1e-4):
    """One kv head of a GQA layer (compress_qk.py:320-382): score the RoPE pairs over the group's query heads
    (K statistics with `ridge_lambda`, Q with sqrt_M's default), keep rank/2 pairs in score order, append the kept
    rows of K and of every query head of the group, and the mask.  Same kernel as compress_layer, one head at a time."""
    q0 = kv_head_idx * kv_head_ratio
    cq = torch.stack([c.to(device=local_device(), dtype=dtype_p) for c in cov_q_layer[q0:q0 + kv_head_ratio]])
    ck = cov_k_layer[kv_head_idx].to(device=local_device(), dtype=dtype_p)[None]
    mask, _, _ = ops.qk_select(cq, ck, rank, _lib.MDG_QK_ROPE_GROUPED, _SQRT_M_DEFAULT_RIDGE, ridge_lambda)
    rows = mask[0]
    K_heads_out.append(_kept_rows(Wk_heads[kv_head_idx], rows.to(Wk_heads.device), slice_dims))
    for Q_head in Wq_heads[q0:q0 + kv_head_ratio]:
        Q_heads_out.append(_kept_rows(Q_head, rows.to(Q_head.device), slice_dims))
    layer_rotary_mask.append(rows)


@torch.no_grad()
def compress_head_llama(C_q: Tensor, C_k: Tensor, Q_head: Tensor, K_head: Tensor, Q_heads_out: list, K_heads_out: list,
                        layer_rotary_mask: list, rank: int, slice_dims=True):
    """One head of an MHA Llama layer (compress_qk.py:387-436): default ridges for both statistics; the kept rows go
    to the lists as CPU bf16 tensors, as upstream."""
    cq = C_q.to(device=local_device(), dtype=dtype_p)[None]
    ck = C_k.to(device=local_device(), dtype=dtype_p)[None]
    mask, _, _ = ops.qk_select(cq, ck, rank, _lib.MDG_QK_ROPE_MHA, _SQRT_M_DEFAULT_RIDGE, _SQRT_M_DEFAULT_RIDGE)
    rows = mask[0]
    Q_heads_out.append(_kept_rows(Q_head, rows.to(Q_head.device), slice_dims).to(device="cpu", dtype=torch.bfloat16))
    K_heads_out.append(_kept_rows(K_head, rows.to(K_head.device), slice_dims).to(device="cpu", dtype=torch.bfloat16))
    layer_rotary_mask.append(rows)


@torch.no_grad()
def compress_head_opt(C_q: Tensor, C_k: Tensor, Q_head: Tensor, K_head: Tensor, bias_Q_head: Tensor, bias_K_head: Tensor,
                      out_Q_heads: list, out_K_heads: list, out_Q_bias: list, out_K_bias: list, rank: int):
    """One OPT head (compress_qk.py:439-476): score_j = |sqrt(Cq)[:, j]| * |sqrt(Ck)[:, j]|, top-`rank` columns in score
    order; rows (to the CPU, as upstream) and the matching bias entries are appended."""
    cq = C_q.to(device=local_device(), dtype=dtype_p)[None]
    ck = C_k.to(device=local_device(), dtype=dtype_p)[None]
    mask, _, _ = ops.qk_select(cq, ck, rank, _lib.MDG_QK_OPT, _SQRT_M_DEFAULT_RIDGE, _SQRT_M_DEFAULT_RIDGE)
    rows = mask[0]
    out_Q_heads.append(Q_head[rows.to(Q_head.device)].to(device="cpu"))
    out_K_heads.append(K_head[rows.to(K_head.device)].to(device="cpu"))
    out_Q_bias.append(bias_Q_head[rows.to(bias_Q_head.device)])
    out_K_bias.append(bias_K_head[rows.to(bias_K_head.device)])


def compress_qk_svd(adapter, cov_x, keep_ratios, rank=None, ridge_lambda=1, slice_dims=True):
    """Present upstream (compress_qk.py:16-150) but unreachable there: nothing calls it and its body ends in a call to
    an undefined name (`slice_QK_dims`), so it cannot run.  Not reproduced."""
    raise NotImplementedError("compress_qk_svd is dead code in the reference (calls the undefined slice_QK_dims); "
                              "use compress_qk")


@torch.no_grad()
def compress_qk(adapter: ModelAdapter, cov, keep_ratios, rank=None, slice_dims=True,
                target_layers: Optional[List[int]] = None):
    """compress_qk.py:152-201.  Returns the list of per-layer rotary masks (None when slice_dims is False)."""
    if target_layers is None:
        target_layers = list(range(adapter.n_layers))
    cov_q_list, cov_k_list = cov
    rotary_masks: List[Tensor] = []
    for i in target_layers:
        rank_i = qk_rank_rule(adapter.head_dim, keep_ratios[i], adapter.arch, rank)
        compress_layer(adapter, i, rank_i, cov_q_list=cov_q_list[i], cov_k_list=cov_k_list[i],
                       rotary_masks=rotary_masks, slice_dims=slice_dims)
        logger.info(f"[QK] Layer {i}: compressed to rank {rank_i} per head (CR-score + interpolation)")
    return rotary_masks if slice_dims else None
