"""CPU: the error bound behind the int8 covariance route (cov_i8.hip's i8_route_kernel, modelled by tests/i8_model.py).

For every distribution family of scripts/probes/i8_fuzz.py the modelled P-plane product is formed exactly (integer class sums)
and compared with the fp64 product:  measured entry-wise error <= the Cauchy-Schwarz bound computed from the per-column plane
energies -- entry by entry, for five and six planes -- and the route the bound selects keeps the measured error below 1e-12
while the guaranteed bound stays below 1.1e-11.  Columns with a bulk far below a few massive activations are exactly the ones
the greedy hands to the fp64 column kernel."""
import numpy as np
import pytest
import torch

from tests import i8_model as M

T, N = 4096, 32


def family(kind, gen):
    g, u, w = (torch.randn(T, N, generator=gen) for _ in range(3))
    c = torch.exp(torch.empty(N).uniform_(np.log(0.05), np.log(2.0), generator=gen))
    silu = torch.nn.functional.silu
    x = {"gaussian": g, "relu": torch.relu(g), "silu_gated": silu(g) * u, "gelu_gated": torch.nn.functional.gelu(g) * u,
         "laplace": torch.sign(g) * torch.log(torch.rand(T, N, generator=gen)), "prod2": g * u, "prod3": g * u * w,
         "student_t4": g / torch.sqrt((torch.randn(4, T, N, generator=gen) ** 2).mean(0)), "cubed": g ** 3,
         "sparse": g * (torch.rand(T, N, generator=gen) < 0.05), "wide_exponents": g * torch.pow(2.0, torch.randint(-30, 1, (T, N), generator=gen).float())}[kind]
    return (x * c).to(torch.bfloat16)


KINDS = ["gaussian", "relu", "silu_gated", "gelu_gated", "laplace", "prod2", "prod3", "student_t4", "cubed", "sparse", "wide_exponents"]


def measured(X, d, E, P):
    ref = X.double().numpy().T @ X.double().numpy()
    dd = np.sqrt(np.diag(ref))
    dd = np.where(dd > 0, dd, 1.0)
    return np.abs(M.product(d, E, P) - ref) / (dd[:, None] * dd[None, :])


@pytest.mark.parametrize("kind", KINDS)
def test_bound_holds_entry_by_entry_and_the_route_keeps_1e12(kind):
    gen = torch.Generator().manual_seed(KINDS.index(kind))
    X = family(kind, gen)
    d, E, N_, rounded, nnz = M.digits(X)
    st = M.column_stats(d, rounded, nnz)
    a, rho = M.alphas(st)
    # the integer lower bound on the column norm is a lower bound, and a tight one
    true_norm = np.sqrt((N_.astype(np.float64) ** 2).sum(0))
    q = st["q"].astype(np.float64)
    lb = (np.sqrt(65536.0 * q[0] + 512.0 * q[6] + q[1]) - 0.5 * np.sqrt(st["nnz"])) * 2.0 ** 32
    assert np.all(lb <= true_norm * (1 + 1e-12)) and np.all(lb >= 0.97 * true_norm)
    for P in (5, 6):
        rel = measured(X, d, E, P)
        assert np.all(rel <= M.entry_bound(a, rho, P) * (1 + 1e-9) + 4e-16), (kind, P)     # (4e-16: the fp64 reference's own rounding)
    planes, cols, (sq, x) = M.route(st, jmax=2, tokens=T)       # (2 of 32 columns may leave: the product's 32 are few against its widths too)
    if planes:
        keep = np.ones(N, bool)
        keep[cols] = False
        rel = measured(X, d, E, planes)[np.ix_(keep, keep)]
        assert sq <= M.TAU_SQ and x <= M.tau_x_of(T)
        assert rel.max() <= min(1e-12, sq + x + 4e-16), (kind, planes, cols, rel.max(), sq, x)
    assert planes == {"gaussian": 5, "relu": 5, "silu_gated": 6, "gelu_gated": 6}.get(kind, planes)


@pytest.mark.parametrize("bulk,planes", [("gaussian", 5), ("silu_gated", 6)])
def test_massive_activation_columns_are_the_ones_that_leave(bulk, planes):
    gen = torch.Generator().manual_seed(3)
    X = family(bulk, gen).float()
    cols = [2, 17, 31]
    for i, c in enumerate(cols):                     # bulk 11-13 binades under three spikes
        top = X[:, c].abs().max()
        X[:, c] *= 2.0 ** -(11 + i)
        X[torch.randperm(T, generator=gen)[:3], c] = top * 1.5
    X = X.to(torch.bfloat16)
    d, E, _, rounded, nnz = M.digits(X)
    st = M.column_stats(d, rounded, nnz)
    got_planes, got_cols, (sq, x) = M.route(st, jmax=6, tokens=T)
    assert (got_planes, got_cols) == (planes, cols)
    # six planes alone would NOT have been enough for such a column (the reason it leaves instead of deepening the launch)
    a, rho = M.alphas(st)
    assert M.violation(a.max(1), rho.max(), 6, M.tau_x_of(T)) > 1.0
    keep = np.ones(N, bool)
    keep[cols] = False
    assert measured(X, d, E, planes)[np.ix_(keep, keep)].max() < 1e-12


def test_a_statistic_beyond_the_column_budget_goes_to_the_fp64_kernel():
    gen = torch.Generator().manual_seed(5)
    g, u = torch.randn(T, 64, generator=gen), torch.randn(T, 64, generator=gen)
    X = (g ** 5 * u ** 3).to(torch.bfloat16)
    d, _, _, rounded, nnz = M.digits(X)
    planes, cols, _ = M.route(M.column_stats(d, rounded, nnz), tokens=T)
    assert planes == 0 and cols == []


def test_worst_case_is_attained_by_a_constant_column():
    """The SQ part of the bound is not slack: a column that repeats one value whose digits straddle planes 2 and 3 under one large
    entry puts the same d_3^2 into every token, and the six-plane error on its diagonal entry equals A_3^2 to rounding."""
    X = torch.full((T, 8), 1.0)
    X[:, 0] = 2.0 ** -17 * 1.4921875          # 8-bit significand 0xBF, 17 binades under the column maximum below
    X[0, 0] = 1.0
    X = X.to(torch.bfloat16)
    d, E, _, rounded, nnz = M.digits(X)
    st = M.column_stats(d, rounded, nnz)
    a, rho = M.alphas(st)
    rel = measured(X, d, E, 6)
    assert rel[0, 0] > 0 and 0.5 < rel[0, 0] / M.entry_bound(a, rho, 6)[0, 0] <= 1.0 + 1e-9


def test_tolerance_factor_trades_planes_for_a_looser_but_still_kept_bound():
    """mdg_cov_i8_set_tolerance's factor in the host model: SiLU-gated columns go from six planes to five at x64, the (SQ, X) the
    route reports stay under the scaled thresholds, and the measured error stays under what it reports."""
    gen = torch.Generator().manual_seed(2)
    X = family("silu_gated", gen)
    d, E, _, rounded, nnz = M.digits(X)
    st = M.column_stats(d, rounded, nnz)
    strict = M.route(st, jmax=2, tokens=T)
    loose = M.route(st, jmax=2, tokens=T, tolerance=64.0)
    assert strict[0] == 6 and loose[0] == 5
    sq, x = loose[2]
    assert sq <= 64 * M.TAU_SQ and x <= 64 * M.tau_x_of(T) and (sq > M.TAU_SQ or x > M.tau_x_of(T))
    keep = np.ones(N, bool)
    keep[loose[1]] = False
    assert measured(X, d, E, 5)[np.ix_(keep, keep)].max() <= sq + x + 4e-16


def test_exact_route_decomposition_is_an_identity_in_integer_arithmetic():
    """cov_i8.hip, "the exact route": N = N_d + L with N_d the top three balanced digits and L = d_3 2^16 + d_4 2^8 + d_5, and
        sum_t N_ti N_tj  =  sum_t N_d,ti N_d,tj  +  sum_t L_ti N_tj  +  sum_t N_d,ti L_tj        (nothing dropped)
    in exact integer arithmetic, for data with deep elements of both signs; the remainder kernel's partner value x_d, recomputed from
    the bf16 value as 2^24 q floor(x / (2^24 q) + 8421504 / 2^24), IS N_d (the balanced digits' rounding, ties included); L lies in
    [-8421504, 8355711]; and the event lists the device builds hold exactly the elements with L != 0 (remainder_counts)."""
    torch.manual_seed(3)
    T, n = 700, 64
    g, u = torch.randn(T, n), torch.randn(T, n)
    X = (torch.nn.functional.silu(g) * u * torch.exp(2 * torch.randn(T, 1))).clamp(-60, 60).to(torch.bfloat16)
    X[0] = 64.0
    X[1, 0], X[2, 0] = 64.0 * 129 * 2.0 ** -22, -64.0 * 129 * 2.0 ** -22          # the rounding tie of the low three digits, both signs
    d, E, N, rounded, _ = M.digits(X)
    assert int(rounded.sum()) == 0 and int(N[1, 0]) & 0xFFFFFF == 0x800000
    L = d[3] * 65536 + d[4] * 256 + d[5]
    Nd = N - L
    assert np.array_equal(Nd, (d[0] << 40) + (d[1] << 32) + (d[2] << 24)) and L.min() >= -8421504 and L.max() <= 8355711
    x = X.double().numpy()
    q = np.ldexp(1.0, (E - 148).astype(np.int64))
    assert np.array_equal(np.floor(x / q + 8421504.0 / 16777216.0) * q, np.ldexp(Nd.astype(np.float64), (E - 172).astype(np.int64)))
    as_int = lambda a: a.astype(object)                     # noqa: E731  (Python integers: exact)
    full = as_int(N).T @ as_int(N)
    parts = as_int(Nd).T @ as_int(Nd) + as_int(L).T @ as_int(N) + as_int(Nd).T @ as_int(L)
    assert (full == parts).all()
    assert int((L != 0).sum()) == int(M.remainder_counts(d).sum()) > 0
    r = M.route_of(X)
    assert r["planes"] in (5, 6) and r["exact"] and r["x"] == 0.0 and r["sq"] == M.EXACT_ROUNDING
    assert not M.route_of(X, offer_exact=False)["exact"]
