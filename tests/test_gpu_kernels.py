"""GPU parity tests of the individual C-ABI entry points against the CPU oracle (tests only import oracle/)."""
import numpy as np
import pytest
import torch

from oracle import modegpt_oracle as O
from tests.golden_util import CASES, ROPE_CASES, Case, RopeCase, canon_rows, vo_products
from tests.i8_limits import check_i8_error

pytestmark = pytest.mark.gpu
F64 = torch.float64


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()


@pytest.fixture(scope="module")
def ops(dev):
    from modegpt_amd import ops as _ops
    return _ops


def acts(gen, tokens, feat, dtype=torch.bfloat16):
    z = torch.randn(tokens, feat, generator=gen)
    c = torch.exp(torch.empty(feat).uniform_(np.log(0.05), np.log(2.0), generator=gen))
    return (z * c).to(dtype)


# ---------------------------------------------------------------- covariance
@pytest.mark.parametrize("name", CASES)
def test_cov_golden(ops, dev, name):
    c = Case(name)
    half = c.tokens // 2
    relu = c.arch == "opt"
    sig = {"mlp": torch.zeros(c.d_ff, c.d_ff, dtype=F64, device=dev), "x": torch.zeros(c.d, c.d, dtype=F64, device=dev),
           "q": torch.zeros(c.n_h, c.hd, c.hd, dtype=F64, device=dev),
           "k": torch.zeros(c.n_kv, c.hd, c.hd, dtype=F64, device=dev)}
    for sl in (slice(0, half), slice(half, c.tokens)):
        ops.cov_accum(sig["mlp"], c.act["h"][sl].to(dev), relu=relu)
        ops.cov_accum(sig["x"], c.act["x"][sl].to(dev).view(1, -1, c.d))
        ops.cov_accum(sig["q"], c.act["q"][sl].to(dev), n_heads=c.n_h)
        ops.cov_accum(sig["k"], c.act["k"][sl].to(dev), n_heads=c.n_kv)
    for k in sig:
        ops.cov_finalize(sig[k], 1.0 / (c.n_texts * 2048))
        want = c.f64["sigma_" + k]
        got = sig[k].cpu()
        assert rel(got, want) < 1e-13, (k, rel(got, want))
        assert torch.equal(got, got.transpose(-1, -2)), "finalize must leave an exactly symmetric matrix"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32, torch.float64])
@pytest.mark.parametrize("tokens,feat,heads", [(777, 384, 1), (1024, 256, 1), (300, 200, 1), (515, 128, 3),
                                               (2048, 64, 4), (40, 16, 2), (0, 64, 1), (5000, 1024, 1)])
def test_cov_shapes(ops, dev, dtype, tokens, feat, heads):
    gen = torch.Generator().manual_seed(tokens * 31 + feat)
    x = acts(gen, tokens, feat * heads, dtype)
    sig = torch.zeros(heads, feat, feat, dtype=F64)
    O.cov_accum_heads(sig, x, heads, feat)
    O.cov_accum_heads(sig, x, heads, feat)
    got = torch.zeros(heads, feat, feat, dtype=F64, device=dev)
    xd = x.to(dev)
    ops.cov_accum(got, xd, n_heads=heads)
    ops.cov_accum(got, xd, n_heads=heads)
    ops.cov_finalize(got, 1.0)
    if tokens == 0:
        assert got.abs().max().item() == 0.0
    else:
        assert rel(got, sig) < 1e-13


def test_cov_strided_rows_and_relu(ops, dev):
    gen = torch.Generator().manual_seed(3)
    full = acts(gen, 600, 512)
    x = full[:, 128:384]  # row stride 512, unit column stride, 16-byte aligned offset
    sig = torch.zeros(256, 256, dtype=F64)
    O.cov_accum_tokens_relu(sig, x)
    got = torch.zeros(256, 256, dtype=F64, device=dev)
    ops.cov_accum(got, full.to(dev)[:, 128:384], relu=True)
    ops.cov_finalize(got, 1.0)
    assert rel(got, sig) < 1e-13
    y = full[:, 3:259]  # misaligned column offset -> scalar staging path
    sig2 = torch.zeros(256, 256, dtype=F64)
    O.cov_accum_tokens(sig2, y)
    got2 = torch.zeros(256, 256, dtype=F64, device=dev)
    ops.cov_accum(got2, full.to(dev)[:, 3:259])
    ops.cov_finalize(got2, 1.0)
    assert rel(got2, sig2) < 1e-13


def test_cov_deterministic(ops, dev):
    gen = torch.Generator().manual_seed(9)
    x = acts(gen, 4096, 128 * 8).to(dev)
    outs = []
    for _ in range(2):
        s = torch.zeros(8, 128, 128, dtype=F64, device=dev)
        ops.cov_accum(s, x, n_heads=8)  # split-K path
        outs.append(s.clone())
    assert torch.equal(outs[0], outs[1])


def test_bi_accum(ops, dev):
    gen = torch.Generator().manual_seed(5)
    a = acts(gen, 3 * 50, 192).view(3, 50, 192)
    b = (a.float() + 0.3 * torch.randn(3, 50, 192, generator=gen)).to(torch.bfloat16)
    want = O.bi_score_batch(a, b) * 50  # oracle returns mean over T of the sum over B
    out = torch.zeros(1, dtype=F64, device=dev)
    ops.bi_accum(out, a.to(dev), b.to(dev))
    assert abs(out.item() - want) / abs(want) < 1e-12


# ---------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 130, 77), (64, 300, 512), (1, 1, 1), (257, 129, 16),
                                   (256, 256, 64), (384, 128, 160), (300, 260, 48)])  # interior tiles: vector staging
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm(ops, dev, M, N, K, ta, tb):
    gen = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=gen, dtype=F64)
    B = torch.randn((N, K) if tb else (K, N), generator=gen, dtype=F64)
    C0 = torch.randn(M, N, generator=gen, dtype=F64)
    want = 0.5 * (A.T if ta else A) @ (B.T if tb else B) - 2.0 * C0
    Cd = C0.to(dev)
    ops.gemm(A.to(dev), B.to(dev), Cd, alpha=0.5, beta=-2.0, trans_a=ta, trans_b=tb)
    assert rel(Cd, want) < 1e-13


def test_gemm_bf16_gather(ops, dev):
    gen = torch.Generator().manual_seed(1)
    A = torch.randn(300, 200, generator=gen, dtype=F64)
    rows = torch.randperm(300, generator=gen)[:150].sort().values
    B = torch.randn(90, 200, generator=gen).to(torch.bfloat16)  # used transposed
    want = A[rows] @ B.double().T
    out = torch.empty(150, 90, dtype=F64, device=dev)
    ops.gemm(A.to(dev), B.to(dev), out, trans_b=True, a_rows=rows.to(dev))
    assert rel(out, want) < 1e-13
    outb = torch.empty(150, 90, dtype=torch.bfloat16, device=dev)
    ops.gemm(A.to(dev), B.to(dev), outb, trans_b=True, a_rows=rows.to(dev))
    assert torch.equal(outb.cpu(), want.to(torch.bfloat16))


def test_gemm_fast_path_bf16_and_gather(ops, dev):
    """Interior tiles, 16-byte aligned operands: the vector-staging path with a gathered fp64 A (k-contiguous) and a
    bf16 B in both layouts, fp64 and bf16 outputs."""
    gen = torch.Generator().manual_seed(11)
    A = torch.randn(700, 512, generator=gen, dtype=F64)
    rows = torch.randperm(700, generator=gen)[:256].sort().values
    Bt = torch.randn(384, 512, generator=gen).to(torch.bfloat16)      # used transposed: k-contiguous B
    want = A[rows] @ Bt.double().T
    out = torch.empty(256, 384, dtype=F64, device=dev)
    ops.gemm(A.to(dev), Bt.to(dev), out, trans_b=True, a_rows=rows.to(dev))
    assert rel(out, want) < 1e-13
    Bn = torch.randn(512, 384, generator=gen).to(torch.bfloat16)      # j-contiguous B
    want2 = A[:256] @ Bn.double()
    out2 = torch.empty(256, 384, dtype=torch.bfloat16, device=dev)
    ops.gemm(A[:256].to(dev), Bn.to(dev), out2)
    assert torch.equal(out2.cpu(), want2.to(torch.bfloat16))
    # a ragged last tile row AND column on that path (k-contiguous operands: rows beyond the end re-read the last valid one)
    rows3 = torch.randperm(700, generator=gen)[:300].sort().values      # 300 = 2 x 128 + 44 gathered rows
    Bt3 = torch.randn(200, 512, generator=gen).to(torch.bfloat16)       # 200 = 128 + 72 columns, used transposed
    C3 = torch.randn(300, 200, generator=gen, dtype=F64)
    want4 = A[rows3] @ Bt3.double().T - 1.5 * C3
    out4 = C3.to(dev)
    ops.gemm(A.to(dev), Bt3.to(dev), out4, beta=-1.5, trans_b=True, a_rows=rows3.to(dev))
    assert rel(out4, want4) < 1e-13
    Ab = torch.randn(256, 512, generator=gen).to(torch.bfloat16)      # bf16 A as in the VO stage
    want3 = Ab.double() @ A[:512, :256].contiguous()
    out3 = torch.zeros(256, 256, dtype=F64, device=dev)
    ops.gemm(Ab.to(dev), A[:512, :256].contiguous().to(dev), out3)
    assert rel(out3, want3) < 1e-13


# ---------------------------------------------------------------- Cholesky family
def spd(gen, n, cond_pow=3.0):
    Q, _ = torch.linalg.qr(torch.randn(n, n, generator=gen, dtype=F64))
    lam = torch.logspace(0, -cond_pow, n, dtype=F64)
    return (Q * lam) @ Q.T


@pytest.mark.parametrize("n", [16, 128, 129, 300, 640, 1000, 1024, 1025, 2048, 2500, 4300])   # (1024 / 2048: the outer blocks of potrf / potrs)
def test_potrf_potrs(ops, dev, n):
    gen = torch.Generator().manual_seed(n)
    A = spd(gen, n)
    L = torch.linalg.cholesky(A)
    Ad = A.to(dev).clone()
    inv = ops.potrf_lower(Ad)
    assert rel(torch.tril(Ad), L) < 1e-11
    B = torch.randn(n, 70, generator=gen, dtype=F64)
    X = torch.cholesky_solve(B, L)
    Bd = B.to(dev).clone()
    ops.potrs_lower(Ad, inv, Bd)
    assert rel(Bd, X) < 1e-9


def test_potrf_not_pd(ops, dev):
    A = torch.eye(200, dtype=F64)
    A[150, 150] = -1.0
    with pytest.raises(torch.linalg.LinAlgError):
        ops.potrf_lower(A.to(dev))


def test_deferred_status_reports_what_the_synchronising_call_would_have_raised(ops, dev):
    """ops.DeferredStatus: inside the block the decomposition entry points enqueue and return (no host round trip for the
    Cholesky pivot / Jacobi flag); check() afterwards raises LinAlgError with the pivot of the FIRST failure, good chains pass,
    and outside the block the calls raise at once as before."""
    good = torch.eye(300, dtype=F64, device=dev) * 2.0
    bad = torch.eye(200, dtype=F64)
    bad[150, 150] = -1.0
    bad2 = torch.eye(200, dtype=F64)
    bad2[20, 20] = -3.0
    with ops.DeferredStatus(dev) as st:
        ops.potrf_lower(good.clone())
        ops.ridge_scores(good, 1e-4)
    st.check()
    st.check()                                        # idempotent
    with ops.DeferredStatus(dev) as st:
        ops.potrf_lower(good.clone())
        ops.potrf_lower(bad.to(dev))                  # does not raise here
        ops.potrf_lower(bad2.to(dev))
    with pytest.raises(torch.linalg.LinAlgError, match="order 151"):
        st.check()
    with pytest.raises(torch.linalg.LinAlgError):     # deferred mode has ended with the block
        ops.potrf_lower(bad.to(dev))
    with pytest.raises(RuntimeError, match="already in deferred mode"):
        with ops.DeferredStatus(dev):
            with ops.DeferredStatus(dev):
                pass
    ops.potrf_lower(good.clone())                     # and the thread is back in the normal mode


@pytest.mark.parametrize("n", [64, 160, 384, 704, 1100])
def test_ridge_scores(ops, dev, n):
    gen = torch.Generator().manual_seed(n + 1)
    H = acts(gen, 3 * n, n).double()
    Cm = H.T @ H / (3 * n)
    lam = float(torch.tensor(1e-4, dtype=torch.float32).double())
    want = O.ridge_scores(Cm, 1e-4)
    got = ops.ridge_scores(Cm.to(dev), lam)
    assert rel(got, want) < 1e-9


@pytest.mark.parametrize("n,k", [(160, 112), (14336, 10035), (1000, 1), (1000, 1000), (5, 3)])
def test_select(ops, dev, n, k):
    gen = torch.Generator().manual_seed(n + k)
    s = torch.rand(n, generator=gen, dtype=F64)
    want = O.mlp_select(s, k)
    got = ops.select_smallest_sorted(s.to(dev), k)
    assert torch.equal(got.cpu(), want)


def test_select_ties_lower_index_first(ops, dev):
    s = torch.tensor([3.0, 1.0, 2.0, 1.0, 1.0, 5.0, float("nan"), 0.5], dtype=F64)
    got = ops.select_smallest_sorted(s.to(dev), 3)
    assert got.cpu().tolist() == [1, 3, 7]


@pytest.mark.parametrize("n,correlated", [(384, False), (640, True), (2048 + 77, True)])
def test_ridge_score_sensitivity_bounds_what_a_relative_error_of_sigma_can_do(ops, dev, n, correlated):
    """mdg_ridge_scores' `sens` (the certificate's first-order bound, include/modegpt_hip.h): the same scores to the bit with or
    without it; sens equals its definition (sum_b |X_bj| sum_a |X_ba| sqrt(c_aa))^2 computed on the CPU from inv(chol(C + ridge I));
    and it does what it is for -- for perturbations |E_ab| <= eps sqrt(c_aa c_bb) of either sign pattern (random, and the adversarial
    one for a given column: E = -eps sign(z_j z_j^T) d d^T) the scores of C + E, recomputed in fp64 on the CPU by the oracle, move
    by at most eps sens_j (plus the second-order term)."""
    gen = torch.Generator().manual_seed(n)
    H = acts(gen, 3 * n, n).double()
    if correlated:                       # mix the features: inv(A) gets dense columns, the triangle inequality has something to lose
        H = H @ (torch.eye(n, dtype=F64) + 0.3 * torch.randn(n, n, generator=gen, dtype=F64) / n ** 0.5)
    Cm = H.T @ H / (3 * n)
    lam = float(torch.tensor(1e-4, dtype=torch.float32).double())
    plain = ops.ridge_scores(Cm.to(dev), lam)
    scores, sens = ops.ridge_scores(Cm.to(dev), lam, want_sens=True)
    assert torch.equal(plain, scores)
    scores, sens = scores.cpu(), sens.cpu()
    A = Cm + lam * torch.eye(n, dtype=F64)
    X = torch.linalg.inv(torch.linalg.cholesky(A))
    d = torch.sqrt(torch.diagonal(Cm))
    want_sens = (X.abs().T @ (X.abs() @ d)) ** 2
    assert rel(sens, want_sens) < 1e-10
    Z = torch.linalg.inv(A)
    exact_first_order = (Z.abs().T @ d) ** 2              # (sum_a sqrt(c_aa) |z_aj|)^2 <= sens_j
    assert bool((exact_first_order <= want_sens * (1 + 1e-12)).all())
    eps = 1e-7
    dd = d[:, None] * d[None, :]
    R = torch.rand(n, n, generator=gen, dtype=F64) * 2 - 1
    trials = [eps * torch.tril(R) * dd]
    for j in (0, n // 2, int(torch.argmax(want_sens / scores))):
        zj = Z[:, j]
        trials.append(-eps * torch.sign(zj[:, None] * zj[None, :]) * dd)
    for E in trials:
        E = torch.tril(E) + torch.tril(E, -1).T
        moved = (O.ridge_scores(Cm + E, 1e-4) - O.ridge_scores(Cm, 1e-4)).abs()
        assert bool((moved <= eps * want_sens * (1 + 1e-3) + 1e-12 * scores).all()), (moved / (eps * want_sens)).max().item()
    # the adversarial pattern comes close to the exact first-order bound for its column: the bound is not slack by construction
    zj = Z[:, 0]
    E = -eps * torch.sign(zj[:, None] * zj[None, :]) * dd
    moved0 = (O.ridge_scores(Cm + E, 1e-4) - O.ridge_scores(Cm, 1e-4))[0].abs().item()
    assert moved0 > 0.99 * eps * exact_first_order[0].item()


def test_selection_margin_certifies_or_flags(ops, dev, caplog):
    """mdg_select_margin + the reporting around it (ModelAdapter.report_selection_margins): a selection whose threshold scores
    are well apart is CERTIFIED against the covariance error bound; a constructed near-tie -- two columns whose scores differ by
    2e-13 relatively, straddling the threshold -- is FLAGGED (warning, metrics entry, certified False), although the selection
    itself is still made (lower score first, as the reference's topk would on these exact numbers)."""
    import logging
    from modegpt_amd import engine
    shape = dict(engine.SHAPES["tiny"], n_layers=1)
    n, keep = shape["d_ff"], 0.7
    rank = int(n * keep)
    gen = torch.Generator().manual_seed(5)
    weights = engine.make_layer_weights(shape, 1, dev)

    def run(diag):
        H = torch.randn(4 * n, n, generator=gen, dtype=F64) * 1e-3
        C = H.T @ H / (4 * n)                                # weak correlations between the columns ...
        C = C - torch.diag(torch.diagonal(C)) + torch.diag(diag)     # ... under an exactly prescribed diagonal
        adapter = engine.TensorAdapter(shape, {0: weights})
        adapter.cov_error_eps = 1.1e-11
        from modegpt_amd.compression.compress_mlp import compress_nystrom
        compress_nystrom(adapter=adapter, cov=[C.to(dev)], keep_ratios=[keep], target_layers=[0])
        adapter.check_chains()
        with caplog.at_level(logging.WARNING, logger="MoDeGPT"):
            caplog.clear()
            rep = adapter.report_selection_margins()
        return adapter, rep[0], [r.getMessage() for r in caplog.records], O.ridge_scores(C, 1e-4)

    # scores ~ 1 / (c_jj + ridge): a geometric ladder of diagonal entries -> every neighbouring pair of scores 1 % apart
    ladder = 0.5 * 1.01 ** torch.arange(n, dtype=F64)
    adapter, rep, warnings, sc = run(ladder[torch.randperm(n, generator=gen)])
    assert rep["certified"] and not warnings and rep["scores_at_risk"] == 0
    srt = torch.sort(sc).values
    assert abs(rep["margin"] - ((srt[rank] - srt[rank - 1]) / srt[rank - 1]).item()) < 1e-9
    assert rep["score_bound"] < 1e-9 < rep["margin"] and rep["eps_certifiable"] > 1e-6
    assert adapter.metrics["mlp_selection"]["0"]["certified"] is True
    # the near-tie: the k-th and (k+1)-th largest diagonal entries 2e-13 apart
    near = ladder.clone()
    order = torch.argsort(near, descending=True)          # smallest scores = largest diagonal entries
    a, b = order[rank - 1].item(), order[rank].item()
    near[b] = near[a] * (1 - 2e-13)
    adapter, rep, warnings, sc = run(near)
    assert not rep["certified"] and rep["scores_at_risk"] >= 2 and rep["margin"] < 1e-11
    assert len(warnings) == 1 and "NOT certified" in warnings[0] and "Layer 0" in warnings[0]
    assert adapter.metrics["mlp_selection"]["0"]["certified"] is False
    got_up = adapter.store[(0, "mlp")]["up"]
    assert got_up.shape[0] == rank                         # (the selection is still made)
    # raw entry point: k == n and k == 0 have nothing to separate
    s = torch.rand(50, dtype=F64, device=dev)
    full = ops.decode_margin(ops.select_margin(s, torch.ones_like(s), torch.arange(50, device=dev), 1e-3).cpu().tolist(), 1e-3)
    none = ops.decode_margin(ops.select_margin(s, torch.ones_like(s), torch.arange(0, device=dev), 1e-3).cpu().tolist(), 1e-3)
    assert full["certified"] and none["certified"] and full["margin"] == float("inf")


def test_gather_rows(ops, dev):
    gen = torch.Generator().manual_seed(2)
    W = torch.randn(500, 264, generator=gen).to(torch.bfloat16)
    rows = torch.randperm(500, generator=gen)[:123]
    assert torch.equal(ops.gather_rows(W.to(dev), rows.to(dev)).cpu(), W[rows])
    W2 = torch.randn(50, 37, generator=gen).to(torch.bfloat16)  # odd width -> scalar path
    assert torch.equal(ops.gather_rows(W2.to(dev), rows[:20].to(dev) % 50).cpu(), W2[rows[:20] % 50])


@pytest.mark.parametrize("name", CASES)
def test_mlp_golden(ops, dev, name):
    c = Case(name)
    Cm = c.f64["sigma_mlp"].to(dev)
    lam32 = float(torch.tensor(c.ridges["nystrom_ridge"], dtype=torch.float32).double())
    scores = ops.ridge_scores(Cm, lam32)
    assert rel(scores, c.f64["mlp_scores"]) < 1e-9
    idx = ops.select_smallest_sorted(scores, c.mlp_rank)
    assert torch.equal(idx.cpu(), c.mlp_idx), "rank selection must be bit-identical"
    assert torch.equal(ops.gather_rows(c.W["up"].to(dev), idx).cpu(), c.bf["mlp_up"])
    down, down64 = ops.nystrom_down(Cm, idx, c.W["down"].to(dev), want_f64=True)
    assert rel(down64, c.f64["mlp_down_f64"]) < 1e-7
    mism = (down.cpu().view(torch.int16) != c.bf["mlp_down"].view(torch.int16)).float().mean().item()
    assert mism < 1e-3, f"bf16 down_proj differs on {mism:.2%} of elements"
    assert rel(down, c.bf["mlp_down"]) < 2 ** -7


# ---------------------------------------------------------------- eigen / QK / VO
@pytest.mark.parametrize("n,batch", [(16, 3), (64, 5), (128, 2), (2, 1)])
def test_syevj(ops, dev, n, batch):
    gen = torch.Generator().manual_seed(n)
    A = torch.randn(batch, n, n, generator=gen, dtype=F64)
    A = A @ A.transpose(1, 2) / n
    lam, V = ops.syevj(A.to(dev))
    lam, V = lam.cpu(), V.cpu()
    want = torch.linalg.eigvalsh(A).flip(-1)
    assert rel(lam, want) < 1e-13
    recon = V @ torch.diag_embed(lam) @ V.transpose(1, 2)
    assert rel(recon, A) < 1e-13
    eye = torch.eye(n, dtype=F64).expand(batch, n, n)
    assert (V.transpose(1, 2) @ V - eye).abs().max().item() < 1e-13


def test_sqrt_psd_small(ops, dev):
    from tests.golden_util import load_misc
    z = load_misc()
    M = torch.from_numpy(z["rd_M"])
    root, inv, _ = ops.sqrt_psd_small(M.to(dev), 1e-5, False, True)
    assert rel(root, torch.from_numpy(z["rd_sqrt"])) < 1e-10
    assert rel(inv, torch.from_numpy(z["rd_invsqrt"])) < 1e-8
    root2, _, _ = ops.sqrt_psd_small(M.to(dev), 1e-3, True, False)
    assert rel(root2, torch.from_numpy(z["rd_sqrt_scaled"])) < 1e-10


@pytest.mark.parametrize("name", CASES)
def test_qk_golden(ops, dev, name):
    from modegpt_amd import _lib
    c = Case(name)
    grouped = c.n_kv != c.n_h
    if c.arch == "opt":
        mode, rq, rk = _lib.MDG_QK_OPT, 1e-4, 1e-4
    elif grouped:
        mode, rq, rk = _lib.MDG_QK_ROPE_GROUPED, 1e-4, c.ridges["ridge_qk"]
    else:
        mode, rq, rk = _lib.MDG_QK_ROPE_MHA, 1e-4, 1e-4
    mask, q_rows, k_rows = ops.qk_select(c.f64["sigma_q"].to(dev), c.f64["sigma_k"].to(dev), c.qk_rank, mode, rq, rk)
    assert torch.equal(mask.cpu(), c.qk_mask), "QK mask (order included) must be bit-identical"
    assert torch.equal(ops.gather_rows(c.W["q"].to(dev), q_rows).cpu(), c.bf["qk_q"])
    assert torch.equal(ops.gather_rows(c.W["k"].to(dev), k_rows).cpu(), c.bf["qk_k"])


@pytest.mark.parametrize("name", CASES)
def test_vo_golden(ops, dev, name):
    c = Case(name)
    v, o, v64, o64 = ops.vo_compress(c.f64["sigma_x"].to(dev), c.W["v"].to(dev), c.W["o"].to(dev), c.n_h, c.n_kv, c.hd,
                                     c.vo_rank, c.ridges["ridge_vo"], want_f64=True)
    r = c.vo_rank
    # invariant: per-head products
    P_got = vo_products(v64.cpu(), o64.cpu(), c.n_h, c.n_kv, r)
    P_ref = vo_products(c.f64["vo_v_f64"], c.f64["vo_o_f64"], c.n_h, c.n_kv, r)
    assert rel(P_got, P_ref) < 1e-8
    # factors up to a per-component sign
    vg, sg = canon_rows(v64.cpu(), r)
    vr, sr = canon_rows(c.f64["vo_v_f64"], r)
    assert rel(vg, vr) < 1e-6
    g = c.n_h // c.n_kv
    for qh in range(c.n_h):
        h = qh // g
        og = o64.cpu()[:, qh * r:(qh + 1) * r] * sg[h * r:(h + 1) * r]
        orf = c.f64["vo_o_f64"][:, qh * r:(qh + 1) * r] * sr[h * r:(h + 1) * r]
        assert rel(og, orf) < 1e-6
    # bf16 outputs are the casts of the fp64 factors
    assert torch.equal(v.cpu(), v64.cpu().to(torch.bfloat16))
    assert torch.equal(o.cpu(), o64.cpu().to(torch.bfloat16))


def test_probe(ops, dev):
    tf = ops.probe_mfma_f64(2048)
    print(f"fp64 MFMA probe: {tf:.1f} TFLOP/s")
    assert 20 < tf < 200


@pytest.mark.parametrize("n", [130, 200, 384, 1024])
def test_sqrt_psd_large(ops, dev, n):
    """sqrt_M beyond head size (block Jacobi) against the oracle's eigh route, incl. a rank-deficient input."""
    gen = torch.Generator().manual_seed(n)
    t = n // 2 if n == 200 else 3 * n          # n == 200: fewer tokens than features -> singular sigma
    X = acts(gen, t, n).double()
    M = X.T @ X / t
    want_s, want_i = O.sqrt_M(M, 1e-5, inverse_sqrt=True)
    root, inv, lam = ops.sqrt_psd_large(M.to(dev), 1e-5, False, True)
    assert rel(root, want_s) < 1e-10
    assert rel(inv, want_i) < 1e-7
    assert rel(lam.sort().values, torch.linalg.eigvalsh(M)) < 1e-11
    want_sc = O.sqrt_M(M, 1e-3, scaled=True)
    root2, _, _ = ops.sqrt_psd_large(M.to(dev), 1e-3, True, False)
    assert rel(root2, want_sc) < 1e-10


@pytest.mark.parametrize("n", [130, 200, 384, 1024])
def test_sqrt_psd_large_newton(ops, dev, n):
    """The GEMM-only route (no eigenvalues requested) against the oracle's eigh route: same sqrt(M + ridge I) and inverse,
    incl. the rank-deficient sigma (n == 200), whose smallest eigenvalue is the ridge itself."""
    gen = torch.Generator().manual_seed(n)
    t = n // 2 if n == 200 else 3 * n
    X = acts(gen, t, n).double()
    M = X.T @ X / t
    want_s, want_i = O.sqrt_M(M, 1e-5, inverse_sqrt=True)
    root, inv, lam = ops.sqrt_psd_large(M.to(dev), 1e-5, False, True, want_evals=False)
    assert lam is None
    assert rel(root, want_s) < 1e-10
    assert rel(inv, want_i) < 1e-7
    assert rel(root @ root, M.to(dev) + 1e-5 * torch.eye(n, dtype=F64, device=dev)) < 1e-11


def test_sqrt_psd_large_newton_falls_back_on_indefinite_input(ops, dev):
    """An eigenvalue below -ridge: there is no real square root of M + ridge I; the reference clamps (sqrt of
    max(lambda + ridge, 0)).  The Newton iteration must notice and hand over to the eigen route."""
    gen = torch.Generator().manual_seed(2)
    Q, _ = torch.linalg.qr(torch.randn(160, 160, generator=gen, dtype=F64))
    lam = torch.linspace(0.01, 3.0, 160, dtype=F64)
    lam[0] = -0.5
    M = (Q * lam) @ Q.T
    M = (M + M.T) / 2
    want = O.sqrt_M(M, 1e-4)
    root, _, _ = ops.sqrt_psd_large(M.to(dev), 1e-4, False, False, want_evals=False)
    assert rel(root, want) < 1e-9


def test_sqrt_M_surface_dispatch(dev):
    """compression_utils.sqrt_M keeps the reference's full domain: batched head-size and d_model-size inputs."""
    from modegpt_amd.compression_utils import sqrt_M
    gen = torch.Generator().manual_seed(4)
    X = acts(gen, 900, 256).double()
    M = X.T @ X / 900
    s, si = sqrt_M(M.to(dev), ridge_lambda=1e-4, inverse_sqrt=True)
    ws, wi = O.sqrt_M(M, 1e-4, inverse_sqrt=True)
    assert rel(s, ws) < 1e-10 and rel(si, wi) < 1e-7
    Mh = torch.stack([M[:64, :64], M[64:128, 64:128]])
    sh = sqrt_M(Mh.to(dev))
    assert rel(sh[1], O.sqrt_M(Mh[1])) < 1e-10


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cov_accum_multi(ops, dev, dtype):
    """Four problems of one batch in one launch == four separate launches == the oracle; incl. the fallback path."""
    gen = torch.Generator().manual_seed(21)
    t = 1000
    H, X, Q, K = acts(gen, t, 512, dtype), acts(gen, t, 256, dtype), acts(gen, t, 4 * 128, dtype), acts(gen, t, 2 * 128, dtype)
    want = [torch.zeros(1, 512, 512, dtype=F64), torch.zeros(1, 256, 256, dtype=F64), torch.zeros(4, 128, 128, dtype=F64),
            torch.zeros(2, 128, 128, dtype=F64)]
    for w, a, h in zip(want, (H, X, Q, K), (1, 1, 4, 2)):
        O.cov_accum_heads(w, a, h, w.shape[-1])
    got = [torch.zeros(512, 512, dtype=F64, device=dev), torch.zeros(256, 256, dtype=F64, device=dev),
           torch.zeros(4, 128, 128, dtype=F64, device=dev), torch.zeros(2, 128, 128, dtype=F64, device=dev)]
    for _ in range(2):
        ops.cov_accum_multi([(got[0], H.to(dev), 1), (got[1], X.to(dev), 1), (got[2], Q.to(dev), 4), (got[3], K.to(dev), 2)])
    for g, w in zip(got, want):
        ops.cov_finalize(g, 0.5)
        assert rel(g.reshape(w.shape), w) < 1e-13
    # fallback: a head size that is not a multiple of 128 cannot be fused
    Q2 = acts(gen, t, 4 * 64, dtype)
    w2 = torch.zeros(4, 64, 64, dtype=F64)
    O.cov_accum_heads(w2, Q2, 4, 64)
    g2, g3 = torch.zeros(4, 64, 64, dtype=F64, device=dev), torch.zeros(256, 256, dtype=F64, device=dev)
    ops.cov_accum_multi([(g3, X.to(dev), 1), (g2, Q2.to(dev), 4)])
    ops.cov_finalize(g2, 1.0)
    assert rel(g2, w2) < 1e-13


# ---------------------------------------------------------------- non-bf16 checkpoints (fp16 OPT, fp32)
@pytest.mark.parametrize("wdt", [torch.float16, torch.float32])
@pytest.mark.parametrize("name", ["tiny_gqa", "tiny_opt"])
def test_weights_not_bf16_are_widened_exactly(ops, dev, name, wdt):
    """The reference widens weights with .to(float64); an fp16 / fp32 weight must not be squeezed through bf16 on its
    way into the factorisations.  The weights here carry mantissa bits bf16 cannot hold."""
    c = Case(name)
    gen = torch.Generator().manual_seed(11)
    jitter = lambda W: (W.float() * (1 + 2 ** -9 * torch.randn(W.shape, generator=gen))).to(wdt)
    Wd, Wv, Wo = jitter(c.W["down"]), jitter(c.W["v"]), jitter(c.W["o"])
    assert not torch.equal(Wd.to(torch.bfloat16).to(wdt), Wd)
    Cm = c.f64["sigma_mlp"]
    ref_down = O.nystrom_down(Cm, Wd, c.mlp_idx)                       # [r, d] fp64
    _, down64 = ops.nystrom_down(Cm.to(dev), c.mlp_idx.to(dev), Wd.to(dev), want_f64=True)
    assert rel(down64, ref_down) < 1e-7
    lossy = O.nystrom_down(Cm, Wd.to(torch.bfloat16), c.mlp_idx)
    assert rel(down64, ref_down) < 1e-3 * rel(lossy, ref_down)

    _, (v_ref, o_ref) = O.compress_vo_layer(Wv, Wo, c.f64["sigma_x"], c.n_h, c.n_kv, c.hd, c.vo_rank, c.ridges["ridge_vo"])
    v, o, v64, o64 = ops.vo_compress(c.f64["sigma_x"].to(dev), Wv.to(dev), Wo.to(dev), c.n_h, c.n_kv, c.hd, c.vo_rank,
                                     c.ridges["ridge_vo"], want_f64=True)
    r = c.vo_rank
    assert rel(vo_products(v64.cpu(), o64.cpu(), c.n_h, c.n_kv, r), vo_products(v_ref, o_ref, c.n_h, c.n_kv, r)) < 1e-8
    assert v.dtype == torch.bfloat16 and o.dtype == torch.bfloat16


# ---------------------------------------------------------------- compressed-model attention
def _ulp_mismatch(got, want):
    """(fraction of elements that differ, max difference in units of the last place of the element dtype)."""
    if got.dtype == torch.float32:
        d = (got.view(torch.int32).long() - want.view(torch.int32).long()).abs()
    else:
        d = (got.view(torch.int16).long() - want.view(torch.int16).long()).abs()
    return (d != 0).float().mean().item(), int(d.max())


def _rope_through_kernel(ops, dev, x_bhtr, cos, sin, mask, n_kv, hd, norm_w=None):
    """x given in the reference's [B, H, T, r] layout; the kernel takes the projection layout [B, T, H*r]."""
    B, H, T, r = x_bhtr.shape
    x = x_bhtr.transpose(1, 2).reshape(B, T, H * r).contiguous().to(dev)
    cs = (cos[:1], sin[:1]) if mask is not None else (cos, sin)      # masked route: batch 0's table (oracle docstring)
    return ops.rope_gather(x, cs[0].to(dev), cs[1].to(dev), None if mask is None else mask.to(dev), H, n_kv, hd,
                           norm_weight=None if norm_w is None else norm_w.to(dev)).cpu()


@pytest.mark.parametrize("name", list(ROPE_CASES))
def test_rope_golden(ops, dev, name):
    """Rotation: bit-identical to the reference's eager expression, for q (grouped heads) and k."""
    c = RopeCase(name)
    assert torch.equal(_rope_through_kernel(ops, dev, c.q, c.cos, c.sin, c.mask, c.n_kv, c.hd), c.q_out)
    assert torch.equal(_rope_through_kernel(ops, dev, c.k, c.cos, c.sin, c.mask, c.n_kv, c.hd), c.k_out)


@pytest.mark.parametrize("name", [n for n in ROPE_CASES if n != "bf16_full"])
def test_rope_masked_norm_golden(ops, dev, name):
    """Qwen3 route: masked RMSNorm then rotation.  The fp32 sum of squares has no defined order in torch, so the
    normalised value may differ in its last place on a few elements; the bar is <= 1 ulp of the dtype on < 1 % of
    elements for the half types; for fp32 (where the rotation's sum can cancel, so ulps of the result say little)
    |diff| <= 4 * 2^-23 * max|x|."""
    c = RopeCase(name)
    nq, nk = c.nq.transpose(1, 2), c.nk.transpose(1, 2)       # the reference's normed tensors, as [B, H, T, r]
    want_q, want_k = O.apply_rotary_compressed(nq, nk, c.cos, c.sin, c.mask)
    got_q = _rope_through_kernel(ops, dev, c.q, c.cos, c.sin, c.mask, c.n_kv, c.hd, c.norm_w)
    got_k = _rope_through_kernel(ops, dev, c.k, c.cos, c.sin, c.mask, c.n_kv, c.hd, c.norm_w)
    for got, want in ((got_q, want_q), (got_k, want_k)):
        frac, ulps = _ulp_mismatch(got, want)
        if c.dtype == torch.float32:
            assert (got - want).abs().max().item() <= 4 * 2 ** -23 * want.abs().max().item(), (frac, ulps)
        else:
            assert ulps <= 1 and frac < 0.01, (frac, ulps)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,T,n_h,n_kv,hd,r,shared", [
    (2, 100, 8, 2, 128, 88, True),     # Llama-3 style, r/2 = 44 -> 8-byte packs, T not a multiple of 16
    (1, 33, 4, 4, 128, 90, True),      # r/2 = 45: scalar packs
    (3, 16, 6, 3, 64, 44, False),      # r/2 = 22: 4-byte packs; per-batch cos/sin on the mask-free route below
    (1, 1, 32, 8, 128, 128, True),     # decode step, nothing dropped
    (2, 7, 2, 1, 256, 256, True),      # widest head the kernel takes
    (1, 5, 2, 2, 16, 2, True),         # one pair kept
    (1, 20, 3, 3, 256, 200, True),     # MHA, wide head: four tokens per thread group, fp32 tables above 64 KB of LDS
])
def test_rope_shapes(ops, dev, dt, B, T, n_h, n_kv, hd, r, shared):
    gen = torch.Generator().manual_seed(B * 1000 + T)
    q = torch.randn(B, n_h, T, r, generator=gen).to(dt)
    ang = torch.rand(B, T, hd // 2, generator=gen) * 6.28
    if shared:
        ang = ang[:1].expand(B, T, hd // 2)
    emb = torch.cat((ang, ang), -1)
    cos, sin = emb.cos().to(dt), emb.sin().to(dt)
    idx = torch.stack([torch.randperm(hd // 2, generator=gen)[:r // 2] for _ in range(n_kv)])
    mask = torch.cat((idx, idx + hd // 2), dim=1)
    kv_like = q[:, :n_kv]
    want_q, want_k = O.apply_rotary_compressed(q, kv_like, cos, sin, mask)
    assert torch.equal(_rope_through_kernel(ops, dev, q, cos, sin, mask, n_kv, hd), want_q)
    assert torch.equal(_rope_through_kernel(ops, dev, kv_like, cos, sin, mask, n_kv, hd), want_k)
    if r == hd:   # mask-free route honours per-batch tables
        want_q, _ = O.apply_rotary_compressed(q, kv_like, cos, sin, None)
        assert torch.equal(_rope_through_kernel(ops, dev, q, cos, sin, None, n_kv, hd), want_q)


def test_rope_strided_input_and_errors(ops, dev):
    """q/k/v often come out of one fused projection: the kernel reads a column slice of a wider row in place."""
    gen = torch.Generator().manual_seed(3)
    B, T, n_h, n_kv, hd, r = 2, 40, 4, 2, 32, 24
    wide = torch.randn(B, T, n_h * r + 50, generator=gen).to(torch.bfloat16).to(dev)
    x = wide[:, :, 10:10 + n_h * r]                       # 2-byte aligned only
    emb = torch.rand(1, T, hd, generator=gen)
    cos, sin = emb.cos().to(torch.bfloat16), emb.sin().to(torch.bfloat16)
    idx = torch.stack([torch.randperm(hd // 2, generator=gen)[:r // 2] for _ in range(n_kv)])
    mask = torch.cat((idx, idx + hd // 2), dim=1)
    got = ops.rope_gather(x, cos.to(dev), sin.to(dev), mask.to(dev), n_h, n_kv, hd).cpu()
    q = x.cpu().view(B, T, n_h, r).transpose(1, 2)
    want, _ = O.apply_rotary_compressed(q, q[:, :n_kv], cos, sin, mask)
    assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.rope_gather(x, cos.to(dev), sin.to(dev), mask[:, :-2].to(dev), n_h, n_kv, hd)
    with pytest.raises(RuntimeError, match="even"):
        ops.rope_gather(wide[:, :, :n_h * 3], cos.to(dev), sin.to(dev), None, n_h, n_kv, hd)
    with pytest.raises(RuntimeError, match="CPU tensor"):
        ops.rope_gather(x.cpu(), cos, sin, mask, n_h, n_kv, hd)


# ---------------------------------------------------------------- int8 digit-plane covariance
def entry_err(S, ref):
    """max over the lower triangle of |S - ref|_ij / sqrt(ref_ii ref_jj)"""
    S, ref = S.double().cpu(), ref.double().cpu()
    d = torch.sqrt(torch.diagonal(ref))
    d = torch.where(d > 0, d, torch.ones_like(d))
    low = torch.tril(torch.ones_like(ref, dtype=torch.bool))
    return (((S - ref).abs() / (d[:, None] * d[None]))[low]).max().item()


def check_route(info, X, tolerance=1.0):
    """The device's route decision against the host model (tests/i8_model.py) on the same bf16 data: planes, the columns handed
    to the fp64 column kernel IN THE ORDER the greedy took them, whether the exact route replaced the truncated product (every
    remainder list fits), and both parts of the bound."""
    from modegpt_amd import ops as _ops
    from tests import i8_model as M
    want = M.route_of(X.cpu(), tolerance=tolerance, offer_exact=_ops.I8_EXACT)
    assert (info["planes"], info["columns"], info["exact"]) == (want["planes"], want["columns"], want["exact"]), \
        (info, {k: want[k] for k in ("planes", "columns", "sq", "x", "exact")})
    if want["planes"]:
        assert abs(info["sq"] - want["sq"]) <= 1e-9 * want["sq"] + 1e-300 and abs(info["x"] - want["x"]) <= 1e-9 * want["x"] + 1e-300
    return want


@pytest.fixture(params=["exact", "truncated"])
def i8_route(request, monkeypatch):
    """Both products behind the int8 covariance: the exact route (MDG_I8_EXACT_ALWAYS: nine plane pairs + the fp64 remainder products
    wherever the remainder lists fit; the default takes it for the six-plane class only) and the truncated five- / six-plane product
    with its bound (MDG_I8_NO_EXACT)."""
    from modegpt_amd import ops as _ops
    monkeypatch.setattr(_ops, "I8_EXACT", request.param == "exact")
    return request.param


def test_cov_i8_tolerance_is_an_argument_of_the_call(ops, dev, monkeypatch):
    """ABI 9: the route's tolerance factor travels with each call (mdg_cov_accum_i8's `tolerance`); the library keeps no accuracy
    state.  SiLU-gated columns need six planes at factor 1 and take five at 64 (fewer plane pairs, a looser but still COMPUTED and
    respected bound); the device's decision equals the host model's at that factor.  Two host threads with different factors, calling
    at the same time, each get the route of their OWN factor -- through the explicit argument and through the thread's default
    (ops.i8_tolerance_scope); the process default (ops.set_i8_tolerance) is Python-side only and untouched by either."""
    import threading
    monkeypatch.setattr(ops, "I8_EXACT", False)       # (the truncated product: on the exact route the factor changes nothing but the class)
    gen = torch.Generator().manual_seed(31)
    T, n = 24576, 256
    X = (torch.nn.functional.silu(torch.randn(T, n, generator=gen)) * torch.randn(T, n, generator=gen)).to(torch.bfloat16)
    ref = torch.zeros(n, n, dtype=F64)
    O.cov_accum_tokens(ref, X)
    Xd = X.to(dev)
    assert ops.i8_tolerance() == 1.0
    got, failures = {}, []
    barrier = threading.Barrier(2)

    def worker(factor, explicit):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for rep in range(3):
                    S = torch.zeros(n, n, dtype=F64, device=dev)
                    info = {}
                    barrier.wait(timeout=60)      # both threads enqueue their call at the same moment
                    if explicit:
                        planes = ops.cov_accum_i8(S, Xd, route_info=info, tolerance=factor)
                    else:
                        with ops.i8_tolerance_scope(factor):
                            assert ops.i8_tolerance() == factor
                            planes = ops.cov_accum_i8(S, Xd, route_info=info)
                    check_route(info, X, tolerance=factor)
                    err = entry_err(S, ref)
                    assert info["sq"] <= factor * 1e-12 and info["x"] <= factor * 1e-11, (factor, info)
                    check_i8_error(err, info["bound"], family="silu_gated", tolerance=factor, ctx=(factor, explicit, rep))
                    got.setdefault(factor, set()).add(planes)
        except BaseException as e:     # noqa: BLE001  (reported by the main thread)
            failures.append((factor, repr(e)))
            barrier.abort()

    for explicit in (True, False):
        threads = [threading.Thread(target=worker, args=(f, explicit)) for f in (1.0, 64.0)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        barrier.reset()
        assert not failures, failures
    assert got == {1.0: {6}, 64.0: {5}}, got
    assert ops.i8_tolerance() == 1.0            # a scope is the thread's own
    # the process default: Python-side, validated, restored
    assert ops.set_i8_tolerance(64.0) == 1.0
    try:
        S = torch.zeros(n, n, dtype=F64, device=dev)
        assert ops.cov_accum_i8(S, Xd) == 5 and ops.cov_accum_i8(S, Xd, tolerance=1.0) == 6
    finally:
        assert ops.set_i8_tolerance(1.0) == 64.0
    with pytest.raises(ValueError, match="outside"):
        ops.set_i8_tolerance(0.5)
    with pytest.raises(RuntimeError, match="outside"):       # the library refuses it too (a caller of the C ABI has no Python check)
        from modegpt_amd import _lib
        import ctypes as C
        lib = _lib.load()
        ws = torch.empty(lib.mdg_cov_accum_i8_ws_bytes(T, n), dtype=torch.uint8, device=dev)
        S = torch.zeros(n, n, dtype=F64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mdg_cov_accum_i8(Xd.data_ptr(), T, n, n, S.data_ptr(), n, ws.data_ptr(), ws.numel(), 0.5, 0, None, None, None, None,
                                            torch.cuda.current_stream(dev).cuda_stream), "mdg_cov_accum_i8")


@pytest.mark.parametrize("tokens,feat", [(777, 256), (4096, 128), (33, 384), (20000, 256), (65504 + 3000, 128)])
def test_cov_i8_matches_the_fp64_oracle(ops, dev, tokens, feat, i8_route):
    """The int8 route (error-free split, truncated product) against the oracle's fp64 X^T X, entry-wise over sqrt(s_ii s_jj):
    below the bound the call itself computed (guaranteed) and below 1e-12 (the empirical figure of this family); the route equals the host model's; a second call accumulates;
    68504 tokens cross the int32 fold boundary (2047 k-steps = 65504 tokens, scripts/probes/i8_int32_bound.py)."""
    gen = torch.Generator().manual_seed(tokens + feat)
    X = acts(gen, tokens, feat)
    ref = torch.zeros(feat, feat, dtype=F64)
    O.cov_accum_tokens(ref, X)
    S = torch.zeros(feat, feat, dtype=F64, device=dev)
    info = {}
    assert ops.cov_accum_i8(S, X.to(dev), route_info=info) in (5, 6)
    check_route(info, X)
    check_i8_error(entry_err(S, ref), info["bound"], family="gaussian", ctx=info)
    X2 = acts(gen, 200, feat)
    O.cov_accum_tokens(ref, X2)
    assert ops.cov_accum_i8(S, X2.to(dev)) in (5, 6)
    check_i8_error(entry_err(S, ref), family="gaussian")      # (two calls: each within its own bound of the accumulated diagonal)


def test_cov_i8_agrees_with_the_fp64_kernel_and_is_deterministic(ops, dev, i8_route):
    gen = torch.Generator().manual_seed(5)
    X = acts(gen, 5000, 512).to(dev)
    S8 = torch.zeros(512, 512, dtype=F64, device=dev)
    S8b = torch.zeros_like(S8)
    S64 = torch.zeros_like(S8)
    assert ops.cov_accum_i8(S8, X) == 5 and ops.cov_accum_i8(S8b, X) == 5
    ops.cov_accum(S64, X)
    assert torch.equal(S8, S8b), "integer accumulation: bit-identical from run to run"
    check_i8_error(entry_err(S8, torch.tril(S64) + torch.tril(S64, -1).T), family="gaussian")
    ops.cov_finalize(S8, 1.0 / 5000)
    assert torch.equal(S8, S8.T)


def test_cov_i8_hands_outlier_columns_to_the_fp64_column_kernel(ops, dev, i8_route):
    """Columns whose bulk sits far below a few massive activations (the BOS-token dimensions of a Llama residual stream) leave
    the int8 path ALONE: the launch stays on five planes, the fold skips their rows and columns, and the fp64 column kernel
    computes those -- plain fp64 arithmetic, so they agree with the oracle to rounding."""
    gen = torch.Generator().manual_seed(6)
    X = acts(gen, 3000, 256)
    X[17, 40] = 3000.0
    X[[5, 900, 2001], 131] = torch.tensor([-2.0e4, 1.5e4, 3.0e4]).to(torch.bfloat16)
    ref = torch.zeros(256, 256, dtype=F64)
    O.cov_accum_tokens(ref, X)
    S = torch.zeros(256, 256, dtype=F64, device=dev)
    info = {}
    ops.i8_route_counts(dev, reset=True)
    assert ops.cov_accum_i8(S, X.to(dev), route_info=info) == 5
    assert sorted(info["columns"]) == [40, 131]
    check_route(info, X)
    assert ops.i8_route_counts(dev, reset=True) == {"i8_5": 1, "i8_6": 0, "fallback_f64": 0, "fp64_columns": 2, "exact": int(i8_route == "exact")}
    assert info["exact"] == (i8_route == "exact")       # (with the two columns gone the remainder lists are nearly empty)
    check_i8_error(entry_err(S, ref), info["bound"], family="outliers", ctx=info)
    low = torch.tril(S).cpu()
    full = low + torch.tril(low, -1).T
    for j in (40, 131):      # the column kernel's entries: fp64 sums of exact products
        assert ((full[j] - ref[j]).abs() / (torch.sqrt(torch.diagonal(ref)) * torch.sqrt(ref[j, j]))).max().item() < 1e-13
    ref_first = ref.clone()
    # a second batch accumulates on top, again with its own columns
    X2 = acts(gen, 1000, 256)
    X2[3, 200] = -1.0e4
    O.cov_accum_tokens(ref, X2)
    info2 = {}
    assert ops.cov_accum_i8(S, X2.to(dev), route_info=info2) == 5 and info2["columns"][0] == 200
    check_route(info2, X2)      # (1000 tokens: the threshold on the cross terms is 1e-12, a few more columns leave to stay on five planes)
    check_i8_error(entry_err(S, ref), max(info["bound"], info2["bound"]), family="outliers", ctx=(info, info2))
    # the same through the strided / unaligned element-wise passes
    wide = torch.zeros(3000, 263, dtype=torch.bfloat16)
    wide[:, 3:259] = X
    S3 = torch.zeros(256, 256, dtype=F64, device=dev)
    assert ops.cov_accum_i8(S3, wide.to(dev)[:, 3:259]) == 5
    check_i8_error(entry_err(S3, ref_first), info["bound"], family="outliers")


@pytest.mark.parametrize("kind", ["silu_gated", "laplace", "relu", "cubed", "student_t"])
def test_cov_i8_route_follows_the_error_bound(ops, dev, kind, i8_route):
    """The route is derived from the per-call bound (Cauchy-Schwarz on the plane energies): light tails -> five planes,
    SiLU-gated products (the MLP statistic of a real Llama) -> six, heavier tails -> columns leave for the fp64 column kernel
    or, when 32 are not enough, the whole statistic goes to the fp64 kernel.  Whatever is chosen must equal the host model's
    choice and hold the measured error below the computed bound (<= 1.1e-11: the guarantee) -- and, on these measured families,
    below the empirical 1e-12."""
    gen = torch.Generator().manual_seed(21)
    T, n = 6000, 256
    g, u = torch.randn(T, n, generator=gen), torch.randn(T, n, generator=gen)
    X = {"silu_gated": torch.nn.functional.silu(g) * u, "laplace": torch.sign(g) * torch.log(torch.rand(T, n, generator=gen)),
         "relu": torch.relu(g), "cubed": g ** 3, "student_t": g / torch.sqrt((torch.randn(3, T, n, generator=gen) ** 2).mean(0))}[kind].to(torch.bfloat16)
    ref = torch.zeros(n, n, dtype=F64)
    O.cov_accum_tokens(ref, X)
    S = torch.zeros(n, n, dtype=F64, device=dev)
    info = {}
    planes = ops.cov_accum_i8(S, X.to(dev), route_info=info)
    want = check_route(info, X)
    assert planes == want["planes"]
    assert planes == {"silu_gated": 6, "relu": 5}.get(kind, planes)
    if i8_route == "exact" and kind in ("silu_gated", "laplace", "relu"):
        assert info["exact"] and info["bound"] < 1e-14, info        # sparse remainders: the exact route, nothing but rounding left
    if i8_route == "truncated":
        assert not info["exact"]
    err = entry_err(S, ref)
    if planes:
        check_i8_error(err, info["bound"], family={"relu": "one_signed"}.get(kind, kind), ctx=info)
    else:
        assert err < 1e-13, err      # the whole statistic went through the fp64 kernel: fp64 rounding only


def test_cov_i8_special_values_and_errors(ops, dev):
    """Zeros, denormals, an all-zero column, tokens not a multiple of the k-step; wrong shapes are refused."""
    gen = torch.Generator().manual_seed(8)
    X = acts(gen, 100, 128)
    X[:, 5] = 0
    X[::3, 9] = torch.tensor(2.0 ** -130, dtype=torch.float32).to(torch.bfloat16)   # bf16 denormals
    X[1::3, 9] = torch.tensor(-2.0 ** -128, dtype=torch.float32).to(torch.bfloat16)
    X[2::3, 9] = 0
    ref = torch.zeros(128, 128, dtype=F64)
    O.cov_accum_tokens(ref, X)
    S = torch.zeros(128, 128, dtype=F64, device=dev)
    ops.cov_accum_i8(S, X.to(dev))
    low = torch.tril(torch.ones(128, 128, dtype=torch.bool))
    check_i8_error(((S.cpu() - ref)[low].abs().max() / ref.abs().max()).item(), family="gaussian")
    assert S[9, 9].item() == ref[9, 9].item() != 0.0          # products of denormals are exact
    Xn = X.clone()
    Xn[3, 7] = float("inf")
    Xn[40, 100] = float("nan")
    Sn = torch.zeros(128, 128, dtype=F64, device=dev)
    info = {}
    assert ops.cov_accum_i8(Sn, Xn.to(dev), route_info=info) in (5, 6)   # Inf / NaN columns go to the fp64 column kernel ...
    assert info["columns"][:2] == [7, 100]
    refn = torch.zeros(128, 128, dtype=F64)
    O.cov_accum_tokens(refn, Xn)
    lown = torch.tril(torch.ones(128, 128, dtype=torch.bool))
    assert torch.equal(torch.isfinite(Sn.cpu())[lown], torch.isfinite(refn)[lown])   # ... which propagates them as the reference does
    assert not bool(torch.isfinite(Sn[7, 7])) and not bool(torch.isfinite(Sn[100, 7]))
    fin = torch.isfinite(refn) & lown
    check_i8_error(((Sn.cpu() - refn)[fin].abs().max() / ref.abs().max()).item(), info["bound"], family="gaussian")
    # An Inf in the LAST token of a call whose token chunks do not end on the column kernel's batch of 8 (1003 tokens -> chunks of
    # 16, the last one 11 long): the entries of that column are +-Inf, as in the reference's fp64 product -- not NaN (the kernel's
    # padding slots used to re-read the last token against a zero: 0 * Inf)
    Xi = acts(gen, 1003, 128).abs() + torch.tensor(0.25).to(torch.bfloat16)       # (no zeros: every reference entry of the column is +Inf)
    Xi[1002, 21] = float("inf")
    Si = torch.zeros(128, 128, dtype=F64, device=dev)
    infoi = {}
    assert ops.cov_accum_i8(Si, Xi.to(dev), route_info=infoi) in (5, 6) and infoi["columns"] == [21]
    refi = torch.zeros(128, 128, dtype=F64)
    O.cov_accum_tokens(refi, Xi)
    got_i, low_i = Si.cpu(), torch.tril(torch.ones(128, 128, dtype=torch.bool))
    assert bool(torch.isinf(refi[21]).all()) and not bool(torch.isnan(got_i[low_i]).any())
    assert torch.equal(torch.isinf(got_i)[low_i], torch.isinf(refi)[low_i]) and bool((got_i[low_i][torch.isinf(got_i)[low_i]] > 0).all())
    with pytest.raises(RuntimeError, match="multiple of 128"):
        ops.cov_accum_i8(torch.zeros(200, 200, dtype=F64, device=dev), acts(gen, 64, 200).to(dev))
    with pytest.raises(ValueError):
        ops.cov_accum_i8(S, X.float().to(dev))


def test_cov_i8_reads_a_column_slice_in_place(ops, dev):
    """Hook inputs are often column slices of a fused projection: the int8 route takes the row pitch as is."""
    gen = torch.Generator().manual_seed(12)
    wide = acts(gen, 1500, 640).to(dev)
    X = wide[:, 128:128 + 384]                       # row pitch 640, 384 columns, 256-byte offset
    assert X.stride(0) == 640 and not X.is_contiguous()
    S = torch.zeros(384, 384, dtype=F64, device=dev)
    assert ops.cov_accum_i8(S, X) == 5
    ref = torch.zeros(384, 384, dtype=F64)
    O.cov_accum_tokens(ref, X.cpu())
    low = torch.tril(torch.ones(384, 384, dtype=torch.bool))
    check_i8_error(((S.cpu() - ref)[low].abs().max() / ref.abs().max()).item(), family="gaussian")
    # rows that are not 16-byte addressable (odd pitch, 6-byte offset): the element-wise passes
    odd = acts(gen, 900, 653).to(dev)
    Y = odd[:, 3:3 + 384]
    S2 = torch.zeros(384, 384, dtype=F64, device=dev)
    assert ops.cov_accum_i8(S2, Y) == 5
    ref2 = torch.zeros(384, 384, dtype=F64)
    O.cov_accum_tokens(ref2, Y.cpu())
    check_i8_error(((S2.cpu() - ref2)[low].abs().max() / ref2.abs().max()).item(), family="gaussian")


def test_cov_i8_randomised_shapes_and_scales(ops, dev):
    """Sweep of shapes, column scales across the whole bf16 exponent range (2^-120 .. 2^120), sparse columns, sign patterns:
    whichever route a call takes, it must agree with the fp64 kernel entry-wise within the bound it computed (guaranteed) and, these
    being measured families, within the empirical 1e-12 of sqrt(sigma_ii sigma_jj)."""
    gen = torch.Generator().manual_seed(2024)
    routes = set()
    for trial in range(12):
        T = int(torch.randint(1, 3000, (1,), generator=gen))
        n = 128 * int(torch.randint(1, 5, (1,), generator=gen))
        z = torch.randn(T, n, generator=gen)
        kind = trial % 4
        if kind == 1:
            z = z * (torch.rand(T, n, generator=gen) < 0.3)              # 70 % zeros
        elif kind == 2:
            z = z.abs()                                                   # one-signed
        elif kind == 3:
            z = torch.nn.functional.silu(z) * torch.randn(T, n, generator=gen)
        expo = torch.randint(-120, 121, (n,), generator=gen).double()
        X = (z.double() * torch.pow(torch.tensor(2.0, dtype=F64), expo)).to(torch.bfloat16).to(dev)
        S8 = torch.zeros(n, n, dtype=F64, device=dev)
        S64 = torch.zeros_like(S8)
        info = {}
        routes.add(ops.cov_accum_i8(S8, X, route_info=info))
        ops.cov_accum(S64, X)
        d = torch.sqrt(torch.diag(S64))
        d = torch.where(d > 0, d, torch.ones_like(d))
        low = torch.tril(torch.ones(n, n, dtype=torch.bool, device=dev))
        err = (((S8 - S64).abs() / (d[:, None] * d[None]))[low]).max().item()
        if info["planes"]:
            check_i8_error(err, info["bound"], family=("gaussian", "sparse", "one_signed", "silu_gated")[kind], ctx=(trial, T, n, kind))
        else:
            assert err < 1e-13, (trial, T, n, kind, err)
        assert bool(torch.isfinite(S8[low]).all())
    assert routes & {5, 6}, routes


def _exact_sigma(X):
    """X^T X of a bf16 matrix in EXACT integer arithmetic (tests/i8_model.digits: x = N 2^(E - 172)): list of rows of Fractions."""
    from fractions import Fraction
    from tests import i8_model as M
    d, E, N, rounded, _ = M.digits(X)
    assert int(rounded.sum()) == 0                       # (no element more than 38 binades under its column maximum here)
    hi, lo = N >> 24, N & 0xFFFFFF                       # N = hi 2^24 + lo: the three int64 products below cannot overflow
    assert np.abs(hi).max() < 2 ** 24 and X.shape[0] < 4096
    hh, hl, ll = hi.T @ hi, hi.T @ lo, lo.T @ lo
    n = X.shape[1]
    out = []
    for i in range(n):
        row = []
        for j in range(n):
            v = (int(hh[i, j]) << 48) + ((int(hl[i, j]) + int(hl[j, i])) << 24) + int(ll[i, j])
            row.append(Fraction(v) * Fraction(2) ** int(E[i] + E[j] - 344))
        out.append(row)
    return out


def test_exact_route_against_exact_integer_arithmetic(ops, dev, monkeypatch):
    """The default product of the int8 covariance drops no plane pair (cov_i8.hip, "the exact route"): X^T X = X_d^T X_d (nine plane
    pairs on the matrix cores) + X_lo^T X + X_d^T X_lo (fp64 sums over the listed remainder elements).  Checked against the EXACT
    sum in integer arithmetic -- not against an fp64 reference, which is less accurate than the thing under test: SiLU-gated columns
    with deep elements of both signs, elements whose low 24 bits are exactly the rounding tie of the balanced digits (and the value
    next to it on both sides), a 900-token call and a 3000-token one accumulated on top.  Entry-wise error <= 5e-15 of
    sqrt(sigma_ii sigma_jj) (MDG_I8_EXACT_ROUNDING; the truncated six-plane product is 6e-14 here, the fp64 kernel 1e-13); the
    workspace is handed over poisoned, results are bit-identical from run to run, and the call says it was exact."""
    from fractions import Fraction
    real_ws = ops._ws

    def poisoned(nbytes, device):
        t, p = real_ws(nbytes, device)
        if t is not None:
            t.fill_(0x55)
        return t, p
    monkeypatch.setattr(ops, "_ws", poisoned)
    gen = torch.Generator().manual_seed(77)
    n = 128

    def make(T):
        g, u = torch.randn(T, n, generator=gen), torch.randn(T, n, generator=gen)
        top = 8.0                                                           # every column's maximum: 8 (E = 130), the products clamped under it
        X = (torch.nn.functional.silu(g) * u).clamp(-7.9, 7.9).to(torch.bfloat16)
        X[0, :] = top
        X[5, 3], X[6, 3], X[7, 3] = top * 129 * 2.0 ** -22, -top * 129 * 2.0 ** -22, top * 2.0 ** -30   # N = 129 2^23: low 24 bits 0x800000, the tie
        X[8, 4], X[9, 4] = top * 255 * 2.0 ** -23, -top * 255 * 2.0 ** -23  # just under it
        X[10, 5], X[11, 5] = top * 131 * 2.0 ** -23, -top * 2.0 ** -37      # just over it; nearly the deepest exact element
        X[12, 6], X[13, 6] = top * 255 * 2.0 ** -30, -top * 255 * 2.0 ** -30  # N = 255 2^15: digits (.., 1, -128, -128, -128) -> L = -8421376 < -2^23
        if T > 2200:      # 2 x 90 rows 16 - 17 binades under the column maximum, in two different 2048-token segments: ~200 listed elements per column
            for r0 in (100, 2100):
                deep = (torch.rand(90, n, generator=gen) + 0.5) * torch.sign(torch.randn(90, n, generator=gen)) * top * 2.0 ** -16
                X[r0:r0 + 90] = deep.to(torch.bfloat16)
        return X
    S = torch.zeros(n, n, dtype=F64, device=dev)
    S2 = torch.zeros_like(S)
    total = [[Fraction(0)] * n for _ in range(n)]
    worst = 0.0
    for T in (900, 3000):
        X = make(T)
        info = {}
        assert ops.cov_accum_i8(S, X.to(dev), route_info=info) == 6 and info["exact"] and info["columns"] == [], info
        assert info["x"] == 0.0 and info["bound"] == 5e-15                  # (nothing rounded to an integer: the rho term is zero)
        # 900 tokens: ~6 listed elements per column -> the tile kernel; 3000 tokens: ~200 -> the wide kernels (both against exact arithmetic)
        assert info["remainder"] == ("tiles" if T == 900 else "wide"), info
        ops.cov_accum_i8(S2, X.to(dev))
        ex = _exact_sigma(X)
        total = [[a + b for a, b in zip(ra, rb)] for ra, rb in zip(total, ex)]
        got = S.cpu()
        diag = [float(total[i][i]) ** 0.5 for i in range(n)]
        for i in range(n):
            for j in range(i + 1):
                worst = max(worst, abs(float(Fraction(got[i, j].item()) - total[i][j])) / (diag[i] * diag[j]))
    assert torch.equal(S, S2), "integer class sums, remainder events added in list order: bit-identical from run to run"
    assert worst <= 5e-15, worst
    # the truncated product on the same data, for scale: within ITS bound, and visibly less exact
    monkeypatch.setattr(ops, "I8_EXACT", False)
    St = torch.zeros(n, n, dtype=F64, device=dev)
    info_t = {}
    X = make(3000)
    assert ops.cov_accum_i8(St, X.to(dev), route_info=info_t) == 6 and not info_t["exact"]
    ex = _exact_sigma(X)
    got = St.cpu()
    diag = [float(ex[i][i]) ** 0.5 for i in range(n)]
    worst_t = max(abs(float(Fraction(got[i, j].item()) - ex[i][j])) / (diag[i] * diag[j]) for i in range(n) for j in range(i + 1))
    assert worst_t <= info_t["bound"] and worst_t > 10 * worst, (worst_t, worst, info_t)


def test_exact_route_gives_way_when_a_remainder_list_does_not_fit(ops, dev):
    """The exact route is an optimisation of the route kernel's decision, never a different answer: where one column holds more
    remainder elements in a 2048-token segment than its list takes (128 = 6.2 %) the call runs the truncated product it would have
    run anyway -- same planes, same columns, the bound of that product -- and the host model predicts which."""
    gen = torch.Generator().manual_seed(5)
    T, n = 6144, 256
    for deep_rows, want_exact in ((120, True), (140, False)):
        X = acts(gen, T, n)
        X[2048:2048 + deep_rows, 64:96] = (X[2048:2048 + deep_rows, 64:96].float() * 2.0 ** -20).to(torch.bfloat16)   # deep_rows deep elements per column in ONE segment
        S = torch.zeros(n, n, dtype=F64, device=dev)
        info = {}
        planes = ops.cov_accum_i8(S, X.to(dev), route_info=info)
        want = check_route(info, X)
        assert planes in (5, 6) and info["exact"] == want_exact == want["exact"], (deep_rows, info)
        ref = torch.zeros(n, n, dtype=F64)
        O.cov_accum_tokens(ref, X)
        check_i8_error(entry_err(S, ref), info["bound"], family="gaussian", ctx=(deep_rows, info))
        assert (info["bound"] < 1e-14) == want_exact


def test_cov_accum_multi_side_stream_overlap_changes_nothing(ops, dev, monkeypatch):
    """In "i8" mode the small per-head fp64 problems run on a side stream beside the last int8 problem
    (ops.COV_OVERLAP_SMALL); results must be bit-identical to the one-stream order, call after call, and later work on
    the caller's stream must see them."""
    gen = torch.Generator().manual_seed(77)
    n, nh, hd, T = 2048, 4, 64, 3000
    monkeypatch.setattr(ops, "I8_MIN_FEATURES", 256)
    batches = [(acts(gen, T, n).to(dev), acts(gen, T, nh * hd).to(dev), acts(gen, T, 2 * hd).to(dev)) for _ in range(3)]

    def run(overlap):
        monkeypatch.setattr(ops, "COV_OVERLAP_SMALL", overlap)
        sx = torch.zeros(n, n, dtype=F64, device=dev)
        sq = torch.zeros(nh, hd, hd, dtype=F64, device=dev)
        sk = torch.zeros(2, hd, hd, dtype=F64, device=dev)
        for x, q, k in batches:
            ops.cov_accum_multi([(sx, x, 1), (sq, q, nh), (sk, k, 2)], mode="i8")
        # consumed on the caller's stream right away, without a device-wide synchronize
        return sx.clone(), sq.clone(), sk.clone()

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    ref = torch.zeros(nh, hd, hd, dtype=F64)
    for _, q, _ in batches:
        O.cov_accum_heads(ref, q.cpu(), nh, hd)
    low = torch.tril(torch.ones(hd, hd, dtype=torch.bool))
    assert ((a[1].cpu() - ref)[:, low].abs().max() / ref.abs().max()).item() < 1e-12      # (head_dim 64: the fp64 kernel)


def test_cov_i8_zero_plane_skipping_is_exact(ops, dev, monkeypatch):
    monkeypatch.setattr(ops, "I8_EXACT", False)       # (this is about the TRUNCATED product kernels' piece-mask skipping)
    """The product kernel skips digit planes that are all-zero over a tile panel in a k-step (piece masks written by the
    split pass), and the split pass does not even WRITE all-zero pieces of planes 4 and 5: the workspace is handed over
    poisoned (0x55 in every byte), so a piece that is read without having been written shows up in the result.  Data built so that the plane depth differs between row groups, between k-steps and between the two panels
    of a tile -- blocks of tiny values (deep planes only there), blocks of exact zeros, single deep elements -- must still
    match the fp64 kernel, and the executed-MFMA count must sit below the dense count yet above the three-plane floor."""
    real_ws = ops._ws

    def poisoned(nbytes, device):
        t, p = real_ws(nbytes, device)
        if t is not None:
            t.fill_(0x55)
        return t, p
    monkeypatch.setattr(ops, "_ws", poisoned)
    gen = torch.Generator().manual_seed(31)
    T, n = 4096 + 17, 384
    X = torch.randn(T, n, generator=gen)
    X[:, 40:72] *= 2.0 ** -2                     # a 32-column group 2 binades down...
    X[0, 40:72] = 1.0                            # ...against a column maximum of 1
    X[1024:2048, 128:256] = 0                    # a whole tile panel of zeros over 32 k-steps
    X[2048:2080, 300] = 2.0 ** -20               # a few deep elements: plane 3 / 4 in one piece only
    X[3000, 10] = 2.0 ** -30
    X = X.to(torch.bfloat16).to(dev)
    S8 = torch.zeros(n, n, dtype=F64, device=dev)
    S64 = torch.zeros_like(S8)
    stats = {}
    planes = ops.cov_accum_i8(S8, X, mfma_stats=stats)
    assert planes == 5
    ops.cov_accum(S64, X)
    d = torch.sqrt(torch.diag(S64))
    low = torch.tril(torch.ones(n, n, dtype=torch.bool, device=dev))
    err = (((S8 - S64).abs() / (d[:, None] * d[None]))[low]).max().item()
    check_i8_error(err, family="gaussian")
    pairs, floor_pairs = {5: (15, 9), 6: (21, 15)}[planes]
    assert stats["dense"] == ops.i8_dense_mfma_count(T, n, planes)
    assert stats["dense"] * floor_pairs // pairs <= stats["executed"] < stats["dense"], stats
    # dense data of full depth in every piece: nothing to skip
    Y = (torch.randn(2048, 128, generator=gen) * torch.pow(2.0, -torch.randint(0, 36, (2048, 128), generator=gen).float()))
    Y = Y.to(torch.bfloat16).to(dev)
    S = torch.zeros(128, 128, dtype=F64, device=dev)
    st2 = {}
    if ops.cov_accum_i8(S, Y, mfma_stats=st2):
        assert st2["executed"] == st2["dense"], st2
    # the six-plane route skips too (SiLU-gated columns: plane 5 never, plane 4 rarely present) and stays exact
    g, u = torch.randn(3000, 256, generator=gen), torch.randn(3000, 256, generator=gen)
    Z = (torch.nn.functional.silu(g) * u).to(torch.bfloat16).to(dev)
    S6, R6, st6 = torch.zeros(256, 256, dtype=F64, device=dev), torch.zeros(256, 256, dtype=F64, device=dev), {}
    assert ops.cov_accum_i8(S6, Z, mfma_stats=st6) == 6
    assert st6["dense"] == ops.i8_dense_mfma_count(3000, 256, 6) and st6["dense"] * 15 // 21 <= st6["executed"] < st6["dense"], st6
    ops.cov_accum(R6, Z)
    d6 = torch.sqrt(torch.diag(R6))
    low6 = torch.tril(torch.ones(256, 256, dtype=torch.bool, device=dev))
    check_i8_error((((S6 - R6).abs() / (d6[:, None] * d6[None]))[low6]).max().item(), family="silu_gated")


# ---------------------------------------------------------------- the collective behind the C ABI
def test_allgather_layers_through_the_c_abi(dev):
    """mdg_comm_unique_id / mdg_comm_init / mdg_allgather_layers / mdg_comm_destroy (RCCL resolved with dlopen) on a
    one-rank communicator: the gathered buffer equals the sent one, on the caller's stream; bad arguments are refused with a
    status, not a crash.  (The N > 1 arithmetic of the same step -- padding, record layout, order -- is covered on CPU over gloo
    in tests/test_dist_gloo.py; more than one GPU is the driver's to run.)"""
    import ctypes as C
    from modegpt_amd import _lib
    lib = _lib.load()
    uid = (C.c_char * 128)()
    _lib.check(lib.mdg_comm_unique_id(uid), "mdg_comm_unique_id")
    comm = C.c_void_p(None)
    with torch.cuda.device(dev):
        _lib.check(lib.mdg_comm_init(C.byref(comm), 1, 0, uid), "mdg_comm_init")
        assert comm.value
        send = torch.randint(0, 255, (3 * 1024 * 1024 + 5,), dtype=torch.uint8, device=dev)
        recv = torch.zeros_like(send)
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.mdg_allgather_layers(send.data_ptr(), recv.data_ptr(), send.numel(), comm, st), "mdg_allgather_layers")
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        assert lib.mdg_allgather_layers(None, recv.data_ptr(), 16, comm, st) == _lib.MDG_ERR_BAD_ARG
        assert lib.mdg_comm_init(C.byref(C.c_void_p(None)), 2, 5, uid) == _lib.MDG_ERR_BAD_ARG
        _lib.check(lib.mdg_comm_destroy(comm), "mdg_comm_destroy")
        assert lib.mdg_comm_destroy(None) == _lib.MDG_OK


def test_cov_i8_route_is_chosen_and_counted_on_the_device(ops, dev):
    """The hooks' path (report=False) never asks the host which route a call took: the five-plane product, the six-plane
    product, the fp64 column kernel and the fp64 kernel are all enqueued and the device runs what the bound asks for.  The
    tallies live on the device (ops.i8_route_counts) and the results equal those of the reporting calls bit for bit."""
    gen = torch.Generator().manual_seed(91)
    n, T = 256, 3000
    g, u = torch.randn(T, n, generator=gen), torch.randn(T, n, generator=gen)
    heavy = (g ** 5 * u ** 3).to(torch.bfloat16)          # every column far beyond six planes: the whole statistic leaves
    data = {"i8_5": acts(gen, T, n), "i8_6": (torch.nn.functional.silu(g) * u).to(torch.bfloat16), "fallback_f64": heavy}
    ops.i8_route_counts(dev, reset=True)
    for k, (route, X) in enumerate(data.items()):
        X = X.to(dev)
        S0 = torch.zeros(n, n, dtype=F64, device=dev)
        S1 = torch.zeros_like(S0)
        assert ops.cov_accum_i8(S0, X, report=False) is None          # enqueue only
        planes = ops.cov_accum_i8(S1, X)                               # the same call, reporting
        assert {5: "i8_5", 6: "i8_6", 0: "fallback_f64"}[planes] == route
        assert torch.equal(S0, S1)
        counts = ops.i8_route_counts(dev)
        assert counts[route] == 2 and sum(v for r, v in counts.items() if r not in ("fp64_columns", "exact")) == 2 * (k + 1), counts
        assert counts["exact"] == (2 if k >= 1 else 0)    # (by default the six-plane class runs the exact route: the faster product there)
    S64 = torch.zeros(n, n, dtype=F64, device=dev)
    ops.cov_accum(S64, data["fallback_f64"].to(dev))
    assert torch.equal(S1, S64)                                        # the whole-statistic fallback IS the fp64 kernel
    assert ops.i8_route_counts(dev, reset=True)["fallback_f64"] == 2 and sum(ops.i8_route_counts(dev).values()) == 0
