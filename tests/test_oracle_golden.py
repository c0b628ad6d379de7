"""CPU: the oracle replays every golden vector captured from the imported reference (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import modegpt_oracle as O
from tests.golden_util import CASES, GOLDEN_DIR, ROPE_CASES, Case, RopeCase, load_misc, vo_products

F64 = torch.float64


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300)).item()


@pytest.mark.parametrize("name", CASES)
def test_sigma(name):
    c = Case(name)
    half = c.tokens // 2
    mlp = torch.zeros(c.d_ff, c.d_ff, dtype=F64)
    x = torch.zeros(c.d, c.d, dtype=F64)
    q = torch.zeros(c.n_h, c.hd, c.hd, dtype=F64)
    k = torch.zeros(c.n_kv, c.hd, c.hd, dtype=F64)
    for sl in (slice(0, half), slice(half, c.tokens)):
        (O.cov_accum_tokens_relu if c.arch == "opt" else O.cov_accum_tokens)(mlp, c.act["h"][sl])
        O.cov_accum_tokens(x, c.act["x"][sl])
        O.cov_accum_heads(q, c.act["q"][sl], c.n_h, c.hd)
        O.cov_accum_heads(k, c.act["k"][sl], c.n_kv, c.hd)
    for s in (mlp, x, q, k):
        O.cov_finalize(s, c.n_texts)
    # all four fixtures come out of the reference's own hook code (gen_golden.py runs _input_hook / _make_proj_hook with their
    # literal "cuda" redirected to the CPU): nothing in them is restated
    assert str(np.load(os.path.join(GOLDEN_DIR, name + ".npz"))["meta_restated"]) == ""
    assert rel(mlp, c.f64["sigma_mlp"]) < 1e-14
    assert rel(x, c.f64["sigma_x"]) < 1e-14
    assert rel(q, c.f64["sigma_q"]) < 1e-14
    assert rel(k, c.f64["sigma_k"]) < 1e-14


@pytest.mark.parametrize("name", CASES)
def test_mlp(name):
    c = Case(name)
    out, (idx, down64, scores) = O.compress_mlp_layer(c.W["up"], None if c.arch == "opt" else c.W["gate"], c.W["down"],
                                                      c.f64["sigma_mlp"], c.keep, c.ridges["nystrom_ridge"])
    assert rel(scores, c.f64["mlp_scores"]) < 1e-12
    assert torch.equal(idx, c.mlp_idx) and idx.numel() == c.mlp_rank
    assert torch.equal(out["up"], c.bf["mlp_up"])
    if c.arch != "opt":
        assert torch.equal(out["gate"], c.bf["mlp_gate"])
    assert torch.equal(out["down"], c.bf["mlp_down"])
    assert rel(down64, c.f64["mlp_down_f64"]) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_sqrt_and_qk_and_vo(name):
    c = Case(name)
    s, si = O.sqrt_M(c.f64["sigma_x"], c.ridges["ridge_vo"], inverse_sqrt=True)
    assert rel(s, c.f64["sqrt_x"]) < 1e-12 and rel(si, c.f64["invsqrt_x"]) < 1e-10
    qk, mask = O.compress_qk_layer(c.W["q"], c.W["k"], c.f64["sigma_q"], c.f64["sigma_k"], c.n_h, c.n_kv, c.hd,
                                   c.qk_rank, c.arch, c.ridges["ridge_qk"])
    assert torch.equal(mask, c.qk_mask)
    assert torch.equal(qk["q_proj"], c.bf["qk_q"]) and torch.equal(qk["k_proj"], c.bf["qk_k"])
    vo, (v64, o64) = O.compress_vo_layer(c.W["v"], c.W["o"], c.f64["sigma_x"], c.n_h, c.n_kv, c.hd, c.vo_rank,
                                         c.ridges["ridge_vo"])
    P = vo_products(v64, o64, c.n_h, c.n_kv, c.vo_rank)
    Pr = vo_products(c.f64["vo_v_f64"], c.f64["vo_o_f64"], c.n_h, c.n_kv, c.vo_rank)
    assert rel(P, Pr) < 1e-10


def test_qk_diagonal_identity():
    """||col_j(sqrt(C + rho I))||^2 == C_jj + rho: the identity the HIP qk_select kernel relies on."""
    c = Case("med_gqa")
    for h in range(c.n_kv):
        s = O.sqrt_M(c.f64["sigma_k"][h], 1e-2)
        n2 = torch.norm(s, dim=0) ** 2
        assert rel(n2, torch.diagonal(c.f64["sigma_k"][h]) + 1e-2) < 1e-12


def test_vo_gram_identity():
    """The Gram reformulation of the VO factors (vo.hip header) against the reference's sqrt/inverse route."""
    c = Case("med_gqa")
    C = c.f64["sigma_x"] + c.ridges["ridge_vo"] * torch.eye(c.d, dtype=F64)
    r, g = c.vo_rank, c.n_h // c.n_kv
    for h in range(c.n_kv):
        Wv = c.W["v"][h * c.hd:(h + 1) * c.hd].double()
        lam, V = torch.linalg.eigh(Wv @ C @ Wv.T)
        lam, V = lam.flip(0), V.flip(1)
        S = lam.clamp(min=0).sqrt()
        v_new = (V[:, :r] / S[:r]).T @ Wv
        for j in range(g):
            qh = h * g + j
            Wo = c.W["o"][:, qh * c.hd:(qh + 1) * c.hd].double()
            o_new = Wo @ (V[:, :r] * S[:r])
            want = c.f64["vo_o_f64"][:, qh * r:(qh + 1) * r] @ c.f64["vo_v_f64"][h * r:(h + 1) * r]
            assert rel(o_new @ v_new, want) < 1e-8


def test_misc_allocate_and_ranks():
    z = load_misc()
    for i in range(int(z["alloc_n"])):
        ratio, smooth, cap = z[f"alloc{i}_par"]
        got = O.allocate_global_sparsity(z[f"alloc{i}_bi"].tolist(), float(ratio), float(smooth), float(cap))
        assert got == z[f"alloc{i}_keep"].tolist()  # bit-identical
    archs = ["llama", "qwen3", "opt"]
    for a, hd, keep, rqk, rvo in z["rank_rules"]:
        assert O.qk_rank(int(hd), float(keep), archs[int(a)]) == int(rqk)
        assert O.vo_rank(int(hd), float(keep), archs[int(a)]) == int(rvo)
    s, si = O.sqrt_M(torch.from_numpy(z["rd_M"]), 1e-5, inverse_sqrt=True)
    assert rel(s, torch.from_numpy(z["rd_sqrt"])) < 1e-12
    assert rel(si, torch.from_numpy(z["rd_invsqrt"])) < 1e-9


@pytest.mark.parametrize("name", list(ROPE_CASES))
def test_compressed_rotary_and_masked_norm(name):
    """G8: the oracle's restatement of the compressed checkpoint's attention semantics replays the reference's outputs
    (LlamaRebuild.apply_rotary_pos_emb with a rotary mask, DenseQwenRebuild._masked_rms_norm) bit for bit."""
    c = RopeCase(name)
    q, k = O.apply_rotary_compressed(c.q, c.k, c.cos, c.sin, c.mask)
    assert torch.equal(q, c.q_out) and torch.equal(k, c.k_out)
    if c.has_norm:
        nq = O.masked_rms_norm(c.q.transpose(1, 2), c.norm_w, 1e-6, c.mask, c.n_h // c.n_kv)
        nk = O.masked_rms_norm(c.k.transpose(1, 2), c.norm_w, 1e-6, c.mask, 1)
        assert torch.equal(nq, c.nq) and torch.equal(nk, c.nk)
