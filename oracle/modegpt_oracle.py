"""CPU oracle for the MoDeGPT per-layer compression path.

TEST INFRASTRUCTURE ONLY.  This module is a torch-CPU / fp64 restatement of the
reference's algorithm for the hot path (covariance accumulation -> per-module
factorisation -> compressed-weight build).  It exists to *check* the HIP path.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; nothing under `modegpt_amd/` does (a test enforces that).

Why torch and not plain C: the reference is Python and its arithmetic lives in
torch.linalg (eigh / cholesky / cholesky_inverse / cholesky_solve / svd / inv ->
LAPACK on CPU).  Restating with the same CPU LAPACK entry points gives the
tightest possible pin.

Parity pin: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is instead pinned against the reference itself,
imported on CPU in the build container by `oracle/gen_golden.py`; the vectors
it produced are committed under `tests/golden/` and `tests/test_oracle_golden.py`
replays them (the reference cannot travel to the GPU box).

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

F64 = torch.float64
SEQ_LEN_NORMALISER = 2048  # calibration.py:141 -- hard-coded regardless of real length


# ----------------------------------------------------------------------------
# covariance accumulation (the hooks)
# ----------------------------------------------------------------------------

def cov_accum_tokens(sigma: torch.Tensor, acts: torch.Tensor) -> None:
    """sigma += H^T H with H = acts flattened to [tokens, features], upcast to
    fp64 *before* the product.  LlamaAdapter.py:127-136 (down_proj pre-hook) and
    LlamaAdapter.py:138-147 (input_layernorm hook: sum_b X_b^T X_b is the same
    flattened product)."""
    H = acts.detach().to(F64).reshape(-1, acts.shape[-1])
    sigma += H.T @ H


def cov_accum_tokens_relu(sigma: torch.Tensor, fc1_out: torch.Tensor) -> None:
    """OPT variant: statistics of ReLU(fc1 output).  model_adapter.py:546-554."""
    H = torch.relu(fc1_out.detach().to(F64)).reshape(-1, fc1_out.shape[-1])
    sigma += H.T @ H


def cov_accum_heads(sigma: torch.Tensor, proj_out: torch.Tensor, n_heads: int, head_dim: int) -> None:
    """sigma[h] += P_h^T P_h for the (pre-RoPE) projection output viewed as
    [tokens, n_heads, head_dim].  LlamaAdapter.py:115-125 / model_adapter.py:556-567."""
    P = proj_out.detach().to(F64).reshape(-1, n_heads, head_dim).permute(1, 0, 2)
    sigma += torch.bmm(P.transpose(1, 2), P)


def cov_finalize(sigma: torch.Tensor, n_texts: int) -> None:
    """sigma /= n_texts * 2048.  calibration.py:141-146."""
    sigma /= n_texts * SEQ_LEN_NORMALISER


def bi_score_batch(x_in: torch.Tensor, x_out: torch.Tensor) -> float:
    """One batch's Block-Influence contribution for one layer:
    mean_T( sum_B (1 - cos(x_in, x_out)) ).  calibration.py:118-124."""
    a = x_in.to(F64)
    b = x_out.to(F64)
    return torch.sum(1 - torch.cosine_similarity(a, b, dim=2), dim=0).mean().item()


# ----------------------------------------------------------------------------
# keep-ratio allocation and rank rules
# ----------------------------------------------------------------------------

def allocate_global_sparsity(
    bi_scores: Sequence[float],
    compression_ratio: float,
    smoothing: float = 0.015,
    max_sparsity: float = 0.8,
    invert: bool = False,
) -> List[float]:
    """BI scores -> per-layer keep ratios.  compression_utils.py:79-124.
    Note the fp32 rounding of the scores (torch.tensor of python floats) before
    the fp64 softmax."""
    L = len(bi_scores)
    s = torch.tensor(list(bi_scores)).to(F64)  # float32 first, on purpose
    if invert:
        s = -s
    w = torch.softmax(-s / smoothing, dim=0)
    sp = w * (L * compression_ratio)
    while True:
        over = sp > max_sparsity
        if not over.any():
            break
        excess = (sp[over] - max_sparsity).sum()
        sp[over] = max_sparsity
        free = ~over
        if free.any():
            sp[free] += excess * (w[free] / w[free].sum())
    return (1 - sp).tolist()


def mlp_rank(d_int: int, keep_ratio: float) -> int:
    """compress_mlp.py:37."""
    return int(d_int * keep_ratio)


def qk_rank(head_dim: int, keep_ratio: float, arch: str, rank: Optional[int] = None) -> int:
    """compress_qk.py:176-182."""
    r = int(head_dim * keep_ratio) if rank is None else rank
    r = max(1, min(r, head_dim))
    if arch == "llama" or "qwen" in arch:
        r = r - (r % 2)
        r = max(2, min(r, head_dim))
    return r


def vo_rank(head_dim: int, keep_ratio: float, arch: str) -> int:
    """compress_vo.py:35-41 (no upper clamp)."""
    r = max(1, int(head_dim * keep_ratio))
    if arch == "llama" or "qwen" in arch:
        r = r - (r % 2)
        r = max(2, r)
    return r


# ----------------------------------------------------------------------------
# sqrt_M
# ----------------------------------------------------------------------------

def sqrt_M(M: torch.Tensor, ridge_lambda: float = 1e-4, scaled: bool = False, inverse_sqrt: bool = False):
    """Symmetric square root with an eigenvalue ridge.  compression_utils.py:15-55.
    lambda += ridge * (max lambda if scaled else 1); sqrt(clamp(lambda, 0));
    optional inverse square root with clamp(sqrt, 1e-12)."""
    lam, V = torch.linalg.eigh(M)
    scale = lam.max() if scaled else 1.0
    lam = lam + ridge_lambda * scale
    root = torch.sqrt(lam.clamp(min=0))
    S = V @ torch.diag(root) @ V.T
    if not inverse_sqrt:
        return S.to(M.dtype)
    inv_root = 1.0 / root.clamp(min=1e-12)
    Sinv = V @ torch.diag(inv_root) @ V.T
    return S.to(M.dtype), Sinv.to(M.dtype)


# ----------------------------------------------------------------------------
# MLP: ridge-leverage selection + Nystrom refit of down_proj
# ----------------------------------------------------------------------------

def ridge_scores(C: torch.Tensor, ridge_lambda: float) -> torch.Tensor:
    """diag((C + fl32(lambda) I)^-1) through cholesky + cholesky_inverse.
    compress_mlp.py:13-25.  The eye is float32, so the ridge actually added is
    the fp32 rounding of lambda."""
    C = C.to(F64)
    n = C.shape[0]
    Cr = C + ridge_lambda * torch.eye(n)  # float32 eye -> fl32(lambda) promoted to fp64
    L = torch.linalg.cholesky(Cr)
    return torch.diag(torch.cholesky_inverse(L))


def mlp_select(scores: torch.Tensor, rank: int) -> torch.Tensor:
    """rank smallest scores, returned as ascending indices.  compress_mlp.py:45-47."""
    idx = torch.topk(scores, k=rank, largest=False, dim=0).indices
    idx, _ = torch.sort(idx)
    return idx


def nystrom_down(C: torch.Tensor, W_d: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """W_d' (as [rank, d_model], fp64) = (C_kk + 1e-6 I)^-1 C[k,:] W_d^T via
    cholesky + cholesky_solve.  compress_mlp.py:52-57."""
    C = C.to(F64)
    Wd = W_d.to(F64)
    r = idx.numel()
    Ckk = C[idx][:, idx]
    cross = C[idx, :] @ Wd.T
    L = torch.linalg.cholesky(Ckk + 1e-6 * torch.eye(r, dtype=F64))
    return torch.cholesky_solve(cross, L)


def compress_mlp_layer(
    W_up: torch.Tensor,
    W_gate: Optional[torch.Tensor],
    W_down: torch.Tensor,
    C: torch.Tensor,
    keep_ratio: float,
    ridge_lambda: float,
):
    """compress_weights + the layout handed to save_layer.  compress_mlp.py:28-64,97.
    Returns dict(up [r,d], gate [r,d] or None, down [d,r]) in bf16 plus
    (idx int64, down_f64 [r,d], scores f64)."""
    scores = ridge_scores(C, ridge_lambda)
    rank = mlp_rank(C.shape[0], keep_ratio)
    idx = mlp_select(scores, rank)
    up = W_up.to(F64)[idx, :].to(torch.bfloat16)
    gate = None if W_gate is None else W_gate.to(F64)[idx, :].to(torch.bfloat16)
    down64 = nystrom_down(C, W_down, idx)
    down = down64.to(torch.bfloat16).T
    return {"up": up, "gate": gate, "down": down}, (idx, down64, scores)


# ----------------------------------------------------------------------------
# QK: CR column selection
# ----------------------------------------------------------------------------

def _col_norms(S: torch.Tensor) -> torch.Tensor:
    return torch.norm(S, dim=0)


def qk_scores_rope(C_q_group: Sequence[torch.Tensor], C_k: torch.Tensor, ridge_k: float, ridge_q: float,
                   take_sqrt: bool) -> torch.Tensor:
    """RoPE-pair CR score for one kv head and its query group.
    GQA: compress_qk.py:344-364 (K ridge = config.ridge_qk, Q ridge = sqrt_M default 1e-4,
    score square-rooted).  MHA: compress_qk.py:403-416 (defaults for both, no sqrt)."""
    hd = C_k.shape[0]
    half = hd // 2
    sk = sqrt_M(C_k.to(F64), ridge_lambda=ridge_k)
    nk1, nk2 = _col_norms(sk[..., :half]), _col_norms(sk[..., half:])
    g = torch.zeros(half, dtype=F64)
    for Cq in C_q_group:
        sq = sqrt_M(Cq.to(F64), ridge_lambda=ridge_q)
        nq1, nq2 = _col_norms(sq[..., :half]), _col_norms(sq[..., half:])
        g = g + (nq1 ** 2 * nk1 ** 2 + nq2 ** 2 * nk2 ** 2)
    return torch.sqrt(g) if take_sqrt else g


def qk_mask_rope(score: torch.Tensor, rank: int) -> torch.Tensor:
    """top rank/2 pair indices in score-descending order, then their partners.
    compress_qk.py:366-367 / :418-419.  NOT sorted."""
    half = score.numel()
    top = torch.topk(score, k=rank // 2).indices
    return torch.cat((top, top + half))


def qk_scores_opt(C_q: torch.Tensor, C_k: torch.Tensor) -> torch.Tensor:
    """OPT: ||sqrt(Cq) col|| * ||sqrt(Ck) col||, default ridges.  compress_qk.py:452-461."""
    sq = sqrt_M(C_q.to(F64))
    sk = sqrt_M(C_k.to(F64))
    return torch.linalg.vector_norm(sq, dim=0) * torch.linalg.vector_norm(sk, dim=0)


def compress_qk_layer(
    W_q: torch.Tensor,
    W_k: torch.Tensor,
    cov_q: torch.Tensor,
    cov_k: torch.Tensor,
    n_heads: int,
    n_kv_heads: int,
    head_dim: int,
    rank: int,
    arch: str,
    ridge_qk: float,
):
    """compress_layer restated.  compress_qk.py:208-308.
    Returns ({"q_proj": [n_heads*r, d], "k_proj": [n_kv*r, d]} bf16, mask int64 [n_kv, r])."""
    grouped = n_kv_heads != n_heads
    Wq = W_q.reshape(n_heads, head_dim, -1)
    Wk = W_k.reshape(n_kv_heads, head_dim, -1)
    q_out, k_out, masks = [], [], []
    rope = arch == "llama" or "qwen" in arch
    for h in range(n_kv_heads):
        if rope and grouped:
            g = n_heads // n_kv_heads
            q0 = h * g
            score = qk_scores_rope([cov_q[j] for j in range(q0, q0 + g)], cov_k[h],
                                   ridge_k=ridge_qk, ridge_q=1e-4, take_sqrt=True)
            m = qk_mask_rope(score, rank)
            k_out.append(Wk[h][m, :])
            for j in range(q0, q0 + g):
                q_out.append(Wq[j][m, :])
        elif arch == "llama":
            score = qk_scores_rope([cov_q[h]], cov_k[h], ridge_k=1e-4, ridge_q=1e-4, take_sqrt=False)
            m = qk_mask_rope(score, rank)
            q_out.append(Wq[h][m, :])
            k_out.append(Wk[h][m, :])
        elif arch == "opt":
            score = qk_scores_opt(cov_q[h], cov_k[h])
            m = torch.topk(score, k=rank).indices
            q_out.append(Wq[h][m, :])
            k_out.append(Wk[h][m, :])
        else:
            raise NotImplementedError(arch)
        masks.append(m.to(torch.int64))
    mask = torch.cat(masks).reshape(n_kv_heads, -1)
    out = {
        "q_proj": torch.cat(q_out, dim=0).to(torch.bfloat16),
        "k_proj": torch.cat(k_out, dim=0).to(torch.bfloat16),
    }
    return out, mask


# ----------------------------------------------------------------------------
# VO: SVD
# ----------------------------------------------------------------------------

def vo_head_grouped(W_v_head, W_o_heads, sqrt_C, inv_sqrt_C, rank):
    """GQA.  compress_vo.py:112-159.  Returns (v' [r,d] f64, [o'_j [d,r] f64 ...])."""
    U, S, Vh = torch.linalg.svd(sqrt_C @ W_v_head.to(F64).T, full_matrices=False)
    v_new = (inv_sqrt_C @ U[:, :rank]).T
    Sd = torch.diag(S)
    o_new = [(Sd[:rank, :rank] @ Vh[:rank, :] @ Wo.to(F64).T).T for Wo in W_o_heads]
    return v_new, o_new


def vo_head_mha(W_v_head, W_o_head, sqrt_C, inv_sqrt_C, rank):
    """MHA, two SVDs.  compress_vo.py:162-223."""
    U, S, Vh = torch.linalg.svd(sqrt_C @ W_v_head.to(F64).T, full_matrices=False)
    A = torch.diag(S) @ Vh @ W_o_head.to(F64).T
    U_p, S_p, V_p = torch.linalg.svd(A, full_matrices=True)
    v_new = (inv_sqrt_C @ U @ U_p)[:, :rank].T
    o_new = (torch.diag(S_p)[:rank, :rank] @ V_p[:rank, :]).T
    return v_new, o_new


def compress_vo_layer(W_v, W_o, cov_x, n_heads, n_kv_heads, head_dim, rank, ridge_vo):
    """compress_vo's per-layer body.  compress_vo.py:43-45,55-90.
    Returns ({"v_proj": [n_kv*r, d], "o_proj": [d, n_heads*r]} bf16, (v_f64, o_f64))."""
    C = cov_x.to(F64)
    sC = sqrt_M(C, ridge_lambda=ridge_vo)
    isC = torch.linalg.inv(sC)
    grouped = n_kv_heads != n_heads
    g = n_heads // n_kv_heads
    vs, os_ = [], []
    for h in range(n_kv_heads):
        Wv_h = W_v[h * head_dim:(h + 1) * head_dim, :]
        if grouped:
            Wo_hs = [W_o[:, (h * g + j) * head_dim:(h * g + j + 1) * head_dim] for j in range(g)]
            v_new, o_new = vo_head_grouped(Wv_h, Wo_hs, sC, isC, rank)
            vs.append(v_new)
            os_.extend(o_new)
        else:
            v_new, o_new = vo_head_mha(Wv_h, W_o[:, h * head_dim:(h + 1) * head_dim], sC, isC, rank)
            vs.append(v_new)
            os_.append(o_new)
    v64 = torch.cat(vs, dim=0)
    o64 = torch.cat(os_, dim=1)
    return {"v_proj": v64.to(torch.bfloat16), "o_proj": o64.to(torch.bfloat16)}, (v64, o64)


# ----------------------------------------------------------------------------
# compressed-model attention semantics: what a compressed
# checkpoint's modeling file does with the rotary masks
# ----------------------------------------------------------------------------

def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """cat(-x[half:], x[:half]) along the last axis.  LlamaRebuild.py:120-124."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rotary_compressed(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                            rotary_mask: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """q [B, n_h, T, r], k [B, n_kv, T, r], cos/sin [B, T, hd], rotary_mask int64 [n_kv, r] or None.
    Every head gathers its own columns of cos/sin (query heads share the row of their kv head), then the usual
    x*cos + rotate_half(x)*sin in the tensors' dtype, one rounded op at a time.  LlamaRebuild.py:153-186.
    Quirk kept: on the masked route the gather index has batch extent 1 (LlamaRebuild.py:167-175), so torch.gather
    returns batch 0's cos/sin rows and those are broadcast to every batch; position tables that differ per batch
    (e.g. left-padded position_ids) are therefore NOT honoured there, while the mask-free route uses them."""
    if rotary_mask is None:
        c, s = cos.unsqueeze(1), sin.unsqueeze(1)
        return q * c + rotate_half(q) * s, k * c + rotate_half(k) * s
    mk = rotary_mask
    mq = torch.repeat_interleave(mk, q.shape[1] // k.shape[1], dim=0)
    pick = lambda tab, m: tab[:1, :, m].permute(0, 2, 1, 3)           # [1, T, H, r] -> [1, H, T, r]
    q_out = q * pick(cos, mq) + rotate_half(q) * pick(sin, mq)
    k_out = k * pick(cos, mk) + rotate_half(k) * pick(sin, mk)
    return q_out, k_out


def masked_rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, rotary_mask: torch.Tensor,
                    groups: int) -> torch.Tensor:
    """x [B, T, H, r]: RMSNorm over the r kept columns in fp32, scaled by the norm weight gathered per head through the
    rotary mask (groups = query heads per kv head, 1 for keys); product in fp32, then back to x.dtype.
    DenseQwenRebuild.py:262-286."""
    xf = x.to(torch.float32)
    normed = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    m = torch.repeat_interleave(rotary_mask, groups, dim=0) if groups > 1 else rotary_mask
    return (weight[m][None, None] * normed).to(x.dtype)


# ----------------------------------------------------------------------------
# whole layer (used by bench.py's cpu_baseline leg and the end-to-end tests)
# ----------------------------------------------------------------------------

def compress_layer_all(weights: dict, covs: dict, shape: dict, keep_ratio: float, ridges: dict):
    """mlp -> qk -> vo for one layer, in the fixed order of run_modegpt.py:128-151."""
    arch = shape["arch"]
    mlp, mlp_aux = compress_mlp_layer(weights["up"], weights.get("gate"), weights["down"],
                                      covs["mlp"], keep_ratio, ridges["nystrom_ridge"])
    r_qk = qk_rank(shape["head_dim"], keep_ratio, arch)
    qk, mask = compress_qk_layer(weights["q"], weights["k"], covs["q"], covs["k"], shape["n_heads"],
                                 shape["n_kv_heads"], shape["head_dim"], r_qk, arch, ridges["ridge_qk"])
    r_vo = vo_rank(shape["head_dim"], keep_ratio, arch)
    vo, vo_aux = compress_vo_layer(weights["v"], weights["o"], covs["x"], shape["n_heads"],
                                   shape["n_kv_heads"], shape["head_dim"], r_vo, ridges["ridge_vo"])
    return {"mlp": mlp, "qk": qk, "vo": vo, "mask": mask, "aux": {"mlp": mlp_aux, "vo": vo_aux}}
