"""MLP compression: ridge-leverage column selection + Nystrom refit of down_proj
(reference: src/compression/compress_mlp.py)."""
from __future__ import annotations

import logging

import torch
from torch import Tensor

from .. import ops
from ..adapters.model_adapter import MLPComponents, ModelAdapter
from ..model_utils import dtype_p, local_device
from ._window import over_layers

logger = logging.getLogger("MoDeGPT")


def _fl32(x: float) -> float:
    """The reference adds `ridge * torch.eye(n)` with a float32 eye (compress_mlp.py:18): the value that reaches
    the fp64 matrix is the fp32 rounding of lambda."""
    return float(torch.tensor(x, dtype=torch.float32).to(torch.float64))


def get_ridge_scores(C: Tensor, layer_idx: int, ridge_lambda=1e-2) -> Tensor:
    """diag((C + fl32(lambda) I)^-1)  (compress_mlp.py:13-25) -- blocked Cholesky + triangular inverse on the
    fp64 MFMA, column norms of L^-1 instead of forming the inverse."""
    C = C.to(dtype=dtype_p, device=local_device())
    return ops.ridge_scores(C, _fl32(ridge_lambda))


@torch.no_grad()
def compress_weights(comps: MLPComponents, C: Tensor, keep_ratio: float, layer_idx: int, ridge_lambda: float):
    """compress_mlp.py:28-64.  Returns (W_u'^T [d, r], W_d' [r, d], W_g'^T [d, r] or None, rank), bf16 --
    the same orientation the reference returns (transposed views of the saved layout)."""
    C = C.to(dtype=dtype_p, device=local_device())
    scores = get_ridge_scores(C, layer_idx=layer_idx, ridge_lambda=ridge_lambda)
    rank = int(C.shape[0] * keep_ratio)
    idx = ops.select_smallest_sorted(scores, rank)                    # topk(largest=False) + sort  (:45-47)
    W_u = comps.up_proj.weight.detach().to(device=local_device(), dtype=torch.bfloat16)
    up = ops.gather_rows(W_u, idx)                                    # W_u[topk, :]               (:49)
    gate = None
    if comps.gate_proj is not None:
        W_g = comps.gate_proj.weight.detach().to(device=local_device(), dtype=torch.bfloat16)
        gate = ops.gather_rows(W_g, idx)                              # W_g[topk, :]               (:50)
    W_d = comps.down_proj.weight.detach().to(device=local_device())              # bf16 as is; fp16/fp32 widen exactly to fp64
    down = ops.nystrom_down(C, idx, W_d, eps=1e-6)                    # [d, r] bf16                (:52-62)
    return up.T, down.T, (None if gate is None else gate.T), rank


@torch.no_grad()
def compress_nystrom(adapter: ModelAdapter, cov, keep_ratios, target_layers, ridge_lambda=1e-4):
    """Per-layer driver (compress_mlp.py:67-117).  As upstream, the ridge actually used is
    adapter.config.nystrom_ridge; the `ridge_lambda` argument is ignored."""
    def enqueue(layer_idx):
        # the layer's whole chain (two Cholesky factorisations, selection, gathers, Nystrom solve) enqueues without a host round
        # trip; the not-positive-definite status of both factorisations is read once (adapter.chain_status)
        comps = adapter.get_mlp_components(layer_idx)
        return compress_weights(comps, cov[layer_idx], keep_ratios[layer_idx], layer_idx=layer_idx,
                                ridge_lambda=adapter.config.nystrom_ridge)

    def retire(layer_idx, result):
        up_T, down_T, gate_T, rank = result
        logger.info(f"[MLP] Layer {layer_idx}  compressed to rank {rank}")
        weights = {"up": up_T.T, "down": down_T.T}
        if gate_T is not None:
            weights["gate"] = gate_T.T
        adapter.save_layer(output_dir=adapter.config.temp_storage_dir, suffix="mlp", weights=weights,
                           layer_idx=layer_idx)

    over_layers(adapter, list(target_layers), enqueue, retire)      # (CHAIN_WIDTH layers' chains in flight, artefacts in layer order)
