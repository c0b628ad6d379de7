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
