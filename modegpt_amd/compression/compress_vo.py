"""VO compression: SVD of sqrt(C) W_v^T per kv head (reference: src/compression/compress_vo.py)."""
from __future__ import annotations

import logging
from typing import List, Optional

import torch
from torch import Tensor

from .. import ops
from ..adapters.model_adapter import ModelAdapter
from ..model_utils import d2, dtype_p

logger = logging.getLogger("MoDeGPT")


def vo_rank_rule(head_dim: int, keep_ratio: float, arch: str) -> int:
    """compress_vo.py:35-41 (no upper clamp upstream; capped at head_dim here because a head has no more
    singular directions than that)."""
    r = max(1, int(head_dim * keep_ratio))
    if arch == "llama" or "qwen" in arch:
        r = r - (r % 2)
        r = max(2, r)
    return min(r, head_dim)


@torch.no_grad()
def compress_vo(adapter: ModelAdapter, cov: List[Tensor], keep_ratios=None, slice_dims=True,
                target_layers: Optional[List[int]] = None):
    """compress_vo.py:13-109.  Grouped (GQA) and two-SVD (MHA) variants are chosen by n_kv_heads != n_heads, as
    upstream; both run inside mdg_vo_compress on the head-sized Gram matrix.  Saves {"v_proj": [n_kv*r, d],
    "o_proj": [d, n_heads*r]} bf16."""
    if target_layers is None:
        target_layers = list(range(adapter.n_layers))
    n_heads, head_dim, arch, n_kv = adapter.n_heads, adapter.head_dim, adapter.arch, adapter.n_kv_heads
    for layer in target_layers:
        rank_i = vo_rank_rule(head_dim, keep_ratios[layer], arch)
        C = cov[layer].to(device=d2, dtype=dtype_p)
        try:
            comps = adapter.get_attn_components(layer)
            W_v, W_o = comps.v_proj.weight, comps.o_proj.weight
        except Exception as e:  # same tolerance as compress_vo.py:47-53
            logger.warning(f"[VO] Layer {layer}: cannot access v_proj/o_proj: {e}")
            continue
        V_heads, O_heads = ops.vo_compress(C, W_v.detach().to(d2), W_o.detach().to(d2), n_heads, n_kv, head_dim, rank_i,
                                           adapter.config.ridge_vo)
        adapter.save_layer(output_dir=adapter.config.temp_storage_dir, suffix="vo",
                           weights={"v_proj": V_heads, "o_proj": O_heads}, layer_idx=layer)
        logger.info(f"[VO] Compressed layer {layer} to rank {rank_i} per head")
