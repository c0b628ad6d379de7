"""sqrt_M and allocate_global_sparsity (reference: src/compression_utils.py)."""
from __future__ import annotations

import logging

import torch
from torch import Tensor
from torch.nn.functional import softmax

from . import ops
from .model_utils import dtype_p

logger = logging.getLogger("MoDeGPT")

_MAX_CLAMP_ITERS = 10_000


@torch.no_grad()
def sqrt_M(M: Tensor, ridge_lambda=1e-4, scaled=False, debug: str = "", inverse_sqrt=False):
    """Symmetric square root with an eigenvalue ridge (compression_utils.py:15-55): eigh, lambda += ridge *
    (max lambda if scaled else 1), sqrt(clamp >= 0), V diag V^T; optionally the inverse root with the 1e-12 clamp.
    M may be [n, n] or [batch, n, n].  Head-sized inputs (even n <= 128) run the batched LDS Jacobi solver
    (mdg_sqrt_psd_small).  Anything larger goes through mdg_sqrt_psd_large, one matrix at a time: when no eigenvalue
    is needed (no `debug` report, ridge not scaled by lambda_max) that is a GEMM-only Newton-Schulz iteration for
    sqrt(M + ridge I) -- the clamps above never bind on a PSD input -- else block Jacobi.  On the Newton route the
    reference's "Negative eigenvalues found" notice (round-off noise of a rank-deficient sigma) is not printed.
    The d_model-sized call the reference makes from compress_vo (compress_vo.py:44) is supported but this engine's own
    compress_vo never makes it (DESIGN.md "Identities")."""
    n = M.shape[-1]
    lam = None
    if n <= 128 and n % 2 == 0:
        root, inv_root, lam = ops.sqrt_psd_small(M, ridge_lambda, scaled, inverse_sqrt)
    else:
        mats = M.reshape(-1, n, n)
        want_evals = bool(debug) or bool(scaled)
        parts = [ops.sqrt_psd_large(m, ridge_lambda, scaled, inverse_sqrt, want_evals) for m in mats]
        root = torch.stack([p[0] for p in parts]).reshape(M.shape)
        inv_root = torch.stack([p[1] for p in parts]).reshape(M.shape) if inverse_sqrt else None
        if want_evals:
            lam = torch.stack([p[2].sort(descending=True).values for p in parts])
    if lam is not None and (debug or bool((lam[..., -1] < 0).any())):
        lam_h = lam.reshape(-1, n).cpu()
        for row in lam_h:
            mx, mn = row[0].item(), row[-1].item()
            if debug:
                print(f"{debug} Pre-reg: {mx:.1e} / {mn:.1e} = {mx / (mn + 1e-9):.1e}")
                print(f"{debug} Pre-reg: eigen.mean() = {row.mean().item():.1e}")
            if mn < 0:
                print(f"Warning: Negative eigenvalues found ({mn}). Matrix is not PSD.")
            if debug:
                post = row + ridge_lambda * (mx if scaled else 1.0)
                print(f"{debug} Post-reg: {post[0].item():.1e} / {post[-1].item():.1e} = "
                      f"{post[0].item() / (post[-1].item() + 1e-9):.1e}")
                print(f"{debug} Post-reg eigen.mean() = {post.mean().item():.1e}")
    root = root.to(dtype=M.dtype)
    if not inverse_sqrt:
        return root
    return root, inv_root.to(dtype=M.dtype)


def get_gate_projs(model, layer_idx):
    """(block, up, down, gate or None, family) of one decoder layer by probing the module tree
    (compression_utils.py:58-76): OPT (fc1 / fc2), GPT-2 style (mlp.c_fc / mlp.c_proj), else Llama style."""
    probes = (("opt", lambda m: m.model.decoder.layers[layer_idx], lambda b: (b.fc1, b.fc2, None)),
              ("gpt", lambda m: m.transformer.h[layer_idx], lambda b: (b.mlp.c_fc, b.mlp.c_proj, None)))
    for family, find_block, parts in probes:
        try:
            block = find_block(model)
            up, down, gate = parts(block)
            return block, up, down, gate, family
        except AttributeError:
            continue
    block = model.model.layers[layer_idx]
    return block, block.mlp.up_proj, block.mlp.down_proj, block.mlp.gate_proj, "llama"


def allocate_global_sparsity(bi_scores, compression_ratio: float, smoothing: float = 0.015, max_sparsity: float = 0.8,
                             adapter=None, invert=False):
    """Block-Influence scores -> per-layer keep ratios (compression_utils.py:79-124).  Host arithmetic on purpose:
    the result feeds int(dim * keep) and must be bit-identical, including the fp32 rounding of the scores by
    torch.tensor(list) before the fp64 softmax.

    One deliberate difference: the reference's clamp loop treats entries pinned AT the cap as free again and
    does not terminate for peaked softmax weights (found while generating goldens, oracle/gen_golden.py).  The
    arithmetic per iteration is identical; this version stops with a RuntimeError after 10 000 iterations."""
    if adapter:
        adapter.metrics["smoothing"] = smoothing
    n_layers = len(bi_scores)
    s = torch.tensor(bi_scores).to(dtype_p)
    if invert:
        s = -s
    total_budget = n_layers * compression_ratio
    softmax_weights = softmax(-s / smoothing, dim=0)
    sparsities = softmax_weights * total_budget
    logger.info(f"Max Layer Sparsity: {sparsities.max().item()}, Avg = {sparsities.mean().item()}")
    if adapter:
        adapter.metrics["max_layer_sparsity"] = sparsities.max().item()
    for _ in range(_MAX_CLAMP_ITERS):
        clamped = sparsities > max_sparsity
        if not clamped.any():
            return (1 - sparsities).tolist()
        excess = (sparsities[clamped] - max_sparsity).sum()
        sparsities[clamped] = max_sparsity
        free = ~clamped
        if free.any():
            sparsities[free] += excess * (softmax_weights[free] / softmax_weights[free].sum())
    raise RuntimeError(
        "allocate_global_sparsity: the cap redistribution did not settle (the reference loops forever on this "
        f"input: ratio={compression_ratio}, smoothing={smoothing}, max_sparsity={max_sparsity}); "
        "use a larger smoothing or a lower ratio")
