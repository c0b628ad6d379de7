"""VO compression: SVD of sqrt(C) W_v^T per kv head (reference: src/compression/compress_vo.py)."""
from __future__ import annotations

import logging
from typing import List, Optional

import torch
from torch import Tensor

from .. import ops
from ..adapters.model_adapter import ModelAdapter
from ..model_utils import dtype_p, local_device
from ._window import over_layers

logger = logging.getLogger("MoDeGPT")


def vo_rank_rule(head_dim: int, keep_ratio: float, arch: str) -> int:
    """compress_vo.py:35-41 (no upper clamp upstream; capped at head_dim here because a head has no more
    singular directions than that)."""
    r = max(1, int(head_dim * keep_ratio))
    if arch == "llama" or "qwen" in arch:
        r = r - (r % 2)
        r = max(2, r)
    return min(r, head_dim)


def _regularised_cov(sqrt_C: Tensor) -> Tensor:
    """The per-head functions of the reference receive sqrt(C + ridge I) (and its inverse) from their caller; the kernel
    works from C + ridge I itself: one fp64-MFMA product recovers it."""
    S = sqrt_C.to(device=local_device(), dtype=dtype_p).contiguous()
    C = torch.empty_like(S)
    ops.gemm(S, S, C)
    return C


@torch.no_grad()
def compress_head_grouped(kv_head_idx: int, kv_head_ratio: int, head_dim: int, rank: int, W_v: Tensor, W_o: Tensor,
                          sqrt_C: Tensor, inv_sqrt_C: Tensor, new_heads_V: list, new_heads_O: list, slice_dims=True,
                          arch="opt"):
    """One kv head of a GQA layer (compress_vo.py:112-159): thin SVD of sqrt(C) W_v,h^T = U S Vh; appends
    W_v' = (C^-1/2 U_r)^T  [rank, d] and, for every query head j of the group, W_o'[j] = (S_r Vh_r W_o[:, j]^T)^T
    [d, rank], fp64 (factors are unique up to a sign per component).  `inv_sqrt_C` is accepted for signature
    compatibility; the Gram route does not need it."""
    if kv_head_ratio < 2:
        raise NotImplementedError("a group of one query head is the MHA case: use compress_head")
    r0, c0 = kv_head_idx * head_dim, kv_head_idx * kv_head_ratio * head_dim
    Wv_h = W_v[r0:r0 + head_dim, :].detach().to(local_device())
    Wo_g = W_o[:, c0:c0 + kv_head_ratio * head_dim].detach().to(local_device())
    _, _, v64, o64 = ops.vo_compress(_regularised_cov(sqrt_C), Wv_h, Wo_g, kv_head_ratio, 1, head_dim, rank, 0.0,
                                     want_f64=True)
    if slice_dims:
        new_heads_V.append(v64)
        for j in range(kv_head_ratio):
            new_heads_O.append(o64[:, j * rank:(j + 1) * rank])
        return
    # upstream's slice_dims=False branch writes the zero-padded V through W_v[:, kv_start:kv_end] (compress_vo.py:135),
    # a column slice of a [n_kv*hd, d] matrix -- shape-inconsistent as written; the row slice is what is meant
    Vp = torch.zeros(head_dim, W_v.shape[1], dtype=W_v.dtype, device=W_v.device)
    Vp[:rank] = v64.to(dtype=W_v.dtype, device=W_v.device)
    W_v[r0:r0 + head_dim, :].data.copy_(Vp)
    for j in range(kv_head_ratio):
        Op = torch.zeros(W_o.shape[0], head_dim, dtype=W_o.dtype, device=W_o.device)
        Op[:, :rank] = o64[:, j * rank:(j + 1) * rank].to(dtype=W_o.dtype, device=W_o.device)
        W_o[:, c0 + j * head_dim:c0 + (j + 1) * head_dim].data.copy_(Op)


@torch.no_grad()
def compress_head(head_idx: int, head_dim: int, rank_i: int, W_v: Tensor, W_o: Tensor, sqrt_C: Tensor, inv_sqrt_C: Tensor,
                  new_heads_V: list, new_heads_O: list, slice_dims=True, arch="opt"):
    """One head of an MHA layer (compress_vo.py:162-223): the two-SVD variant; appends W_v' [rank, d] and W_o' [d, rank]
    (fp64), or with slice_dims=False writes them zero-padded into W_v / W_o in place."""
    s0 = head_idx * head_dim
    Wv_h = W_v[s0:s0 + head_dim, :].detach().to(local_device())
    Wo_h = W_o[:, s0:s0 + head_dim].detach().to(local_device())
    _, _, v64, o64 = ops.vo_compress(_regularised_cov(sqrt_C), Wv_h, Wo_h, 1, 1, head_dim, rank_i, 0.0, want_f64=True)
    if slice_dims:
        new_heads_V.append(v64)
        new_heads_O.append(o64)
        return
    Vp = torch.zeros(head_dim, W_v.shape[1], dtype=W_v.dtype, device=W_v.device)
    Vp[:rank_i] = v64.to(dtype=W_v.dtype, device=W_v.device)
    Op = torch.zeros(W_o.shape[0], head_dim, dtype=W_o.dtype, device=W_o.device)
    Op[:, :rank_i] = o64.to(dtype=W_o.dtype, device=W_o.device)
    W_v[s0:s0 + head_dim, :].data.copy_(Vp)
    W_o[:, s0:s0 + head_dim].data.copy_(Op)


@torch.no_grad()
def compress_vo(adapter: ModelAdapter, cov: List[Tensor], keep_ratios=None, slice_dims=True,
                target_layers: Optional[List[int]] = None):
    """compress_vo.py:13-109.  Grouped (GQA) and two-SVD (MHA) variants are chosen by n_kv_heads != n_heads, as
    upstream; both run inside mdg_vo_compress on the head-sized Gram matrix.  Saves {"v_proj": [n_kv*r, d],
    "o_proj": [d, n_heads*r]} bf16."""
    if target_layers is None:
        target_layers = list(range(adapter.n_layers))
    n_heads, head_dim, arch, n_kv = adapter.n_heads, adapter.head_dim, adapter.arch, adapter.n_kv_heads
    ranks = {}

    def enqueue(layer):
        ranks[layer] = rank_i = vo_rank_rule(head_dim, keep_ratios[layer], arch)
        C = cov[layer].to(device=local_device(), dtype=dtype_p)
        try:
            comps = adapter.get_attn_components(layer)
            W_v, W_o = comps.v_proj.weight, comps.o_proj.weight
        except Exception as e:  # same tolerance as compress_vo.py:47-53
            logger.warning(f"[VO] Layer {layer}: cannot access v_proj/o_proj: {e}")
            return None
        # (the eigensolver's convergence flag is read once per layer, by the adapter: over_layers)
        return ops.vo_compress(C, W_v.detach().to(local_device()), W_o.detach().to(local_device()), n_heads, n_kv, head_dim,
                               rank_i, adapter.config.ridge_vo)

    def retire(layer, result):
        if result is None:
            return
        V_heads, O_heads = result
        adapter.save_layer(output_dir=adapter.config.temp_storage_dir, suffix="vo",
                           weights={"v_proj": V_heads, "o_proj": O_heads}, layer_idx=layer)
        logger.info(f"[VO] Compressed layer {layer} to rank {ranks[layer]} per head")

    over_layers(adapter, list(target_layers), enqueue, retire)
