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


I8_GUARANTEED_EPS = 1.1e-11      # mdg_cov_accum_i8: entry-wise over sqrt(sigma_ii sigma_jj), for any input, at tolerance factor 1


def covariance_error_eps(adapter, n_features: int) -> float:
    """The entry-wise relative error bound eps of the sigma_mlp this run accumulated, |sigma_ij - exact| <= eps sqrt(sigma_ii sigma_jj)
    -- what the selection certificate is taken against.  `adapter.cov_error_eps` when the caller states it; else from the route:
    int8 digit planes -> the route's guarantee 1.1e-11 x the tolerance factor (+ one fp64 rounding per fold); fp64 matrix cores ->
    the worst case of an fp64 sum of exact products over the calibration tokens, (tokens / 4 + 4) 2^-53 (v_mfma_f64_16x16x4 adds four
    products per step; typical errors are ~sqrt of that count: 6e-14 at 10^6 tokens).  A run whose statistics took both routes
    (a fallback) gets the larger."""
    stated = getattr(adapter, "cov_error_eps", None)
    if stated is not None:
        return float(stated)
    tokens = int(getattr(adapter, "calib_tokens", 0) or getattr(adapter.config, "calib_size", 32) * 2048)
    eps_f64 = (tokens / 4 + 4) * 2.0 ** -53
    routes = getattr(adapter, "cov_routes", None) or {}
    int8 = ops.COV_MODE == "i8" and n_features % 128 == 0 and n_features >= ops.I8_MIN_FEATURES and adapter.arch != "opt"
    if not int8:
        return eps_f64
    eps_i8 = I8_GUARANTEED_EPS * ops.i8_tolerance() + 64 * 2.0 ** -53
    return max(eps_i8, eps_f64) if routes.get("fallback_f64", 0) or routes.get("fp64_columns", 0) else eps_i8


@torch.no_grad()
def compress_weights(comps: MLPComponents, C: Tensor, keep_ratio: float, layer_idx: int, ridge_lambda: float,
                     margin_eps: float = None, margin_out: list = None):
    """compress_mlp.py:28-64.  Returns (W_u'^T [d, r], W_d' [r, d], W_g'^T [d, r] or None, rank), bf16 --
    the same orientation the reference returns (transposed views of the saved layout).
    margin_eps / margin_out (not upstream): with both given, the certificate of the rank selection against an entry-wise relative
    error margin_eps of C (ops.select_margin: 8 numbers on the device) is appended to margin_out -- two more passes over the
    triangular inverse the scores come from, nothing else changes."""
    C = C.to(dtype=dtype_p, device=local_device())
    rank = int(C.shape[0] * keep_ratio)
    if margin_out is not None and margin_eps is not None:
        scores, sens = ops.ridge_scores(C, _fl32(ridge_lambda), want_sens=True)
        idx = ops.select_smallest_sorted(scores, rank)
        margin_out.append(ops.select_margin(scores, sens, idx, margin_eps))
    else:
        scores = get_ridge_scores(C, layer_idx=layer_idx, ridge_lambda=ridge_lambda)
        idx = ops.select_smallest_sorted(scores, rank)                # topk(largest=False) + sort  (:45-47)
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
        record = getattr(adapter, "selection_margin", None)          # (a duck-typed adapter without it: no certificate)
        if record is None:
            return compress_weights(comps, cov[layer_idx], keep_ratios[layer_idx], layer_idx=layer_idx,
                                    ridge_lambda=adapter.config.nystrom_ridge)
        eps, margin = covariance_error_eps(adapter, cov[layer_idx].shape[0]), []
        result = compress_weights(comps, cov[layer_idx], keep_ratios[layer_idx], layer_idx=layer_idx,
                                  ridge_lambda=adapter.config.nystrom_ridge, margin_eps=eps, margin_out=margin)
        # the selection's certificate stays on the device until the adapter next waits for the chain (report_selection_margins)
        record(layer_idx, margin[0], eps)
        return result

    def retire(layer_idx, result):
        up_T, down_T, gate_T, rank = result
        logger.info(f"[MLP] Layer {layer_idx}  compressed to rank {rank}")
        weights = {"up": up_T.T, "down": down_T.T}
        if gate_T is not None:
            weights["gate"] = gate_T.T
        adapter.save_layer(output_dir=adapter.config.temp_storage_dir, suffix="mlp", weights=weights,
                           layer_idx=layer_idx)

    over_layers(adapter, list(target_layers), enqueue, retire)      # (CHAIN_WIDTH layers' chains in flight, artefacts in layer order)
