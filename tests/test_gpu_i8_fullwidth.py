"""The int8 digit-plane covariance (the DEFAULT route for sigma_mlp / sigma_x, ops.I8_MIN_FEATURES = 2048 and up) checked at the
widths the product actually sends to it: 2048, 4096, 5120 (Qwen3-14B d), 11008 (Llama-2-7B d_ff), 14336 (Llama-3-8B d_ff),
17408 (Qwen3-14B d_ff), one calibration batch of 16 x 2048 tokens, on Gaussian columns (five planes) AND SiLU-gated columns
(six planes: what a real Llama MLP feeds the hook, LlamaAdapter.py:127-136).

What is compared with what:
  * n = 2048, 4096: the CPU oracle's fp64 H^T H (oracle/modegpt_oracle.py:cov_accum_tokens) directly;
  * every width: the v_mfma_f64 kernel (ops.cov_accum, itself held to 1e-13 of the oracle by test_cov_golden / test_cov_shapes
    and spot-checked against torch at full size in test_gpu_e2e.py), ENTRY-WISE over the lower triangle, normalised by
    sqrt(sigma_ii sigma_jj): within the bound the call computed (the guarantee, <= 1.1e-11) and -- both data kinds here being
    families the route was measured on -- within the empirical 1e-12 (tests/i8_limits.py says which is which);
  * every width: spot blocks against a plain torch fp64 product of the same columns + the trace checksum sum(x^2);
  * n = 14336: the MLP rank selection (ridge scores -> k smallest, sorted) from sigma_i8 must be IDENTICAL to the one from
    sigma_f64 (north_star: "rank selections bit-identical"), at keep 0.7 and 0.6 (BASELINE configs 3 and 4).
"""
import math

import pytest
import torch

from oracle import modegpt_oracle as O
from tests.i8_limits import check_i8_error

pytestmark = pytest.mark.gpu
F64 = torch.float64
T_BATCH = 16 * 2048
WIDTHS = [2048, 4096, 5120, 11008, 14336, 17408]


@pytest.fixture(scope="module")
def ops(dev):
    from modegpt_amd import ops as _ops
    return _ops


@pytest.fixture(params=["exact", "truncated"])
def i8_route(request, monkeypatch):
    """Both products behind the int8 covariance: the exact route (MDG_I8_EXACT_ALWAYS: the top three digit planes' nine plane pairs +
    the fp64 remainder products, wherever the remainder lists fit -- they do on both data kinds here; the default takes it for the
    six-plane class only) and the truncated five- / six-plane product with its bound (MDG_I8_NO_EXACT)."""
    from modegpt_amd import ops as _ops
    monkeypatch.setattr(_ops, "I8_EXACT", request.param == "exact")
    return request.param


def gaussian(dev, tokens, feat, seed):
    """bench.py's generator: z * c_j, c_j log-uniform[0.05, 2]."""
    g = torch.Generator(device=dev).manual_seed(seed)
    c = torch.exp(torch.empty(feat, device=dev).uniform_(math.log(0.05), math.log(2.0), generator=g))
    return (torch.randn(tokens, feat, device=dev, generator=g) * c).to(torch.bfloat16)


def silu_gated(dev, tokens, feat, seed):
    """silu(g) * u with per-column scales: the shape of act_fn(gate_proj(x)) * up_proj(x), the input of down_proj."""
    g = torch.Generator(device=dev).manual_seed(seed)
    c = torch.exp(torch.empty(feat, device=dev).uniform_(math.log(0.05), math.log(2.0), generator=g))
    a = torch.nn.functional.silu(torch.randn(tokens, feat, device=dev, generator=g))
    a.mul_(torch.randn(tokens, feat, device=dev, generator=g)).mul_(c)
    return a.to(torch.bfloat16)


def entrywise_err(S, R):
    """max over the lower triangle of |S - R|_ij / sqrt(R_ii R_jj), in row chunks (17408^2 fp64 temporaries are 2.4 GB each)."""
    n = R.shape[0]
    d = torch.sqrt(torch.diagonal(R))
    d = torch.where(d > 0, d, torch.ones_like(d))
    worst = 0.0
    for r0 in range(0, n, 2048):
        r1 = min(n, r0 + 2048)
        e = (S[r0:r1] - R[r0:r1]).abs() / (d[r0:r1, None] * d[None, :])
        rows = torch.arange(r0, r1, device=R.device)[:, None]
        cols = torch.arange(n, device=R.device)[None, :]
        e = torch.where(cols <= rows, e, torch.zeros_like(e))
        worst = max(worst, e.max().item())
    return worst


def spot_and_trace(S, X, step):
    """Blocks of 192 columns (straddling 128-row tile and 2 x 2 super-block borders) against torch's fp64 product -- entry-wise
    over sqrt(sigma_ii sigma_jj) like everything else here -- and the trace."""
    n = S.shape[0]
    starts = sorted({0, 64, n // 2 - 96, n - 192, (n // 128 // 2) * 128 - 64, 3 * step % (n - 192)})
    Xd = {s: X[:, s:s + 192].double() for s in starts}
    nrm = {s: torch.sqrt((Xd[s] ** 2).sum(0)) for s in starts}
    worst = 0.0
    for si in starts:
        for sj in starts:
            if sj > si:
                continue
            want = Xd[si].T @ Xd[sj]
            got = S[si:si + 192, sj:sj + 192]
            if si == sj:
                got, want = torch.tril(got), torch.tril(want)
            elif sj + 192 > si:       # overlapping column ranges: part of the block lies above the diagonal
                rows = torch.arange(si, si + 192, device=S.device)[:, None]
                cols = torch.arange(sj, sj + 192, device=S.device)[None, :]
                keep = cols <= rows
                got, want = got * keep, want * keep
            worst = max(worst, ((got - want).abs() / (nrm[si][:, None] * nrm[sj][None, :])).max().item())
    tr = sum((X[:, c0:c0 + 2048].double() ** 2).sum().item() for c0 in range(0, n, 2048))
    return worst, abs(torch.diagonal(S).sum().item() - tr) / tr


@pytest.mark.parametrize("kind,planes", [("gaussian", 5), ("silu_gated", 6)])
@pytest.mark.parametrize("n", WIDTHS)
def test_i8_route_at_product_widths(ops, dev, n, kind, planes, i8_route):
    X = (gaussian if kind == "gaussian" else silu_gated)(dev, T_BATCH, n, 100 + n)
    S8 = torch.zeros(n, n, dtype=F64, device=dev)
    S64 = torch.zeros_like(S8)
    stats, info = {}, {}
    assert ops.cov_accum_i8(S8, X, mfma_stats=stats, route_info=info) == planes
    assert 0 < stats["executed"] <= stats["dense"]
    assert info["exact"] == (i8_route == "exact"), info
    if info["exact"]:    # the launch of the three top planes: exactly its nine plane pairs, whatever class the data has
        assert stats["planes_run"] == 3 and stats["executed"] == stats["dense"] and info["bound"] < 1e-14, (stats, info)
    ops.cov_accum(S64, X)
    check_i8_error(entrywise_err(S8, S64), info["bound"], family=kind, ctx=(n, info))
    spot, trace = spot_and_trace(S8, X, n)
    check_i8_error(spot, info["bound"], family=kind, ctx=(n, "spot blocks against torch"))
    assert trace < 1e-13, (n, kind, trace)
    if n <= 4096:   # straight against the oracle (CPU fp64; seconds at these widths)
        ref = torch.zeros(n, n, dtype=F64)
        O.cov_accum_tokens(ref, X.cpu())
        check_i8_error(entrywise_err(S8.cpu(), ref), info["bound"], family=kind, ctx=(n, "oracle"))
    # a second batch accumulates on top (the hook's +=), and the mirrored, normalised result is exactly symmetric
    X2 = (gaussian if kind == "gaussian" else silu_gated)(dev, 4096 + 40, n, 7 + n)
    info2 = {}
    assert ops.cov_accum_i8(S8, X2, route_info=info2) in (5, 6)
    ops.cov_accum(S64, X2)
    check_i8_error(entrywise_err(S8, S64), max(info["bound"], info2["bound"]), family=kind, ctx=(n, "two batches"))
    ops.cov_finalize(S8, 1.0 / (T_BATCH + 4136))
    assert torch.equal(S8, S8.T)


@pytest.mark.parametrize("kind,planes", [("gaussian", 5), ("silu_gated", 6)])
def test_i8_route_across_the_int32_fold_with_super_blocks(ops, dev, kind, planes, i8_route):
    """n = 4096 (32 row blocks -> 16 x 16 super-block rows on 8 XCDs) with 65504 + 4000 tokens: the int32 classes are folded
    into sigma once inside the launch and once at its end."""
    n, T = 4096, 65504 + 4000
    X = (gaussian if kind == "gaussian" else silu_gated)(dev, T, n, 11)
    S8 = torch.zeros(n, n, dtype=F64, device=dev)
    S64 = torch.zeros_like(S8)
    info = {}
    assert ops.cov_accum_i8(S8, X, route_info=info) == planes
    ops.cov_accum(S64, X)
    check_i8_error(entrywise_err(S8, S64), info["bound"], family=kind, ctx=info)
    ref = torch.zeros(n, n, dtype=F64)
    O.cov_accum_tokens(ref, X.cpu())
    check_i8_error(entrywise_err(S8.cpu(), ref), info["bound"], family=kind, ctx="oracle")


@pytest.mark.parametrize("kind", ["gaussian", "silu_gated"])
def test_mlp_rank_selection_is_identical_on_both_routes(ops, dev, kind):
    """Llama-3-8B d_ff: ridge scores of sigma_i8 vs sigma_f64 (compress_mlp.py:13-25 with the fp32-rounded ridge of the
    tests.sh recipe) select the SAME index set at keep 0.7 (rank 10035) and keep 0.6 (rank 8601)."""
    n = 14336
    mk = gaussian if kind == "gaussian" else silu_gated
    S8 = torch.zeros(n, n, dtype=F64, device=dev)
    S64 = torch.zeros_like(S8)
    for b in range(2):
        X = mk(dev, T_BATCH, n, 500 + b)
        assert ops.cov_accum_i8(S8, X) in (5, 6)
        ops.cov_accum(S64, X)
        del X
    ops.cov_finalize(S8, 1.0 / (2 * T_BATCH))
    ops.cov_finalize(S64, 1.0 / (2 * T_BATCH))
    lam = float(torch.tensor(1e-4, dtype=torch.float32).double())
    sc8, sc64 = ops.ridge_scores(S8, lam), ops.ridge_scores(S64, lam)
    assert ((sc8 - sc64).abs() / sc64).max().item() < 1e-9
    for keep in (0.7, 0.6):
        r = int(n * keep)
        assert torch.equal(ops.select_smallest_sorted(sc8, r), ops.select_smallest_sorted(sc64, r)), (kind, keep)


@pytest.mark.parametrize("n,T,kind", [(8192, 65504 + 2000, "gaussian"), (8192, 65504 + 2000, "silu_gated"), (8320, 3000, "gaussian"),
                                      (12416, 2100, "silu_gated"), (8192, 33, "gaussian"),
                                      (4096, 100, "gaussian"), (4096, 33, "silu_gated"), (2944, 5000, "gaussian"),
                                      (3328, 5 * 65504 + 3000, "gaussian"), (3712, 5 * 65504 + 3000, "silu_gated")])
def test_persistent_launch_shapes(ops, dev, n, T, kind, i8_route):
    """Statistics of 2048 features and more run as the persistent launch (one workgroup per CU working through static tile
    lists): the int32 fold boundary inside a tile list (65504 tokens), row-block counts that leave ragged groups along the
    diagonal (65 and 97 blocks), and a call shorter than the LDS ring is deep.  The tiles of the last, partly filled round are
    cut into k-chunks that fold into partial tiles (i8_tail_combine_kernel adds them to sigma): 4096 features -> 16 tiles in
    15 chunks (five planes) / 32 tiles in 7 chunks (six), with 4 or 2 k-steps in the whole call most chunks are EMPTY;
    2944 features -> 20 tiles in 11 chunks; 3328 / 3712 features -> 95 / 102 tiles in five chunks each, worked in two short
    rounds, each chunk longer than the int32 fold interval (a partial tile is folded into more than once)."""
    X = (gaussian if kind == "gaussian" else silu_gated)(dev, T, n, 31 + n)
    S8 = torch.zeros(n, n, dtype=F64, device=dev)
    S64 = torch.zeros_like(S8)
    info = {}
    assert ops.cov_accum_i8(S8, X, route_info=info) in (5, 6)
    ops.cov_accum(S64, X)
    check_i8_error(entrywise_err(S8, S64), info["bound"], family=kind, ctx=(n, T, info))
    # every tile of the lower triangle was visited exactly once: a second call doubles the result exactly
    ops.cov_accum_i8(S8, X)
    low = torch.tril(torch.ones(n, n, dtype=torch.bool, device=dev))
    ref = S64 * 2
    check_i8_error(entrywise_err(S8, ref), info["bound"], family=kind, ctx=(n, T, "second call"))
    assert bool(torch.isfinite(S8[low]).all())


def _fused_case(ops, dev, widths, heads, T, kind_first, seed):
    """(sigma_i8 list, sigma_f64 list, route) of one fused call: full statistics of `widths` (the first one of kind
    `kind_first`, the others Gaussian) + per-head statistics `heads` = [n_heads, ...] of head_dim 128."""
    items, refs = [], []
    for i, n in enumerate(widths):
        X = (silu_gated if (i == 0 and kind_first == "silu_gated") else gaussian)(dev, T, n, seed + i)
        items.append((torch.zeros(n, n, dtype=F64, device=dev), X, 1))
    for j, nh in enumerate(heads):
        X = gaussian(dev, T, nh * 128, seed + 10 + j)
        items.append((torch.zeros(nh, 128, 128, dtype=F64, device=dev), X, nh))
    for sigma, X, nh in items:
        R = torch.zeros_like(sigma)
        ops.cov_accum(R, X, n_heads=nh)
        refs.append(R)
    info = []
    route = ops.cov_accum_i8_multi(items, report=True, route_info=info)
    return items, refs, route, info


def _check_against(S, R, bound=None, family="gaussian"):
    """bound: the statistic's own reported bound where the caller read it back (else the guarantee 1.1e-11); family: the data of
    the _fused_case / cov_accum_multi tests are gaussian or silu_gated -- both measured families (empirical 1e-12 asserted too)."""
    if S.dim() == 2:
        check_i8_error(entrywise_err(S, R), bound, family=family)
    else:
        for h in range(S.shape[0]):
            check_i8_error(entrywise_err(S[h], R[h]), bound, family=family, ctx=h)


@pytest.mark.parametrize("widths,heads,T,kind,planes", [
    ([4096, 2048], [8, 2], 12000, "gaussian", 5),           # 528 + 136 + 8 + 2 tiles of 128 x 128, k-split tail across statistics
    ([4096, 2048], [8, 2], 12000, "silu_gated", 6),         # the gated statistic takes every other one to six planes
    ([2048], [32], 700, "gaussian", None),                  # (few tokens: the bound of some column may ask for six planes -> either route)
    ([14336, 4096], [32, 8], 32768, "gaussian", 5),         # the four hooks of a Llama-3-8B layer, one calibration batch
    ([14336, 4096], [32, 8], 32768, "silu_gated", 6),
    ([3328], [3], 2 * 65504 + 100, "gaussian", 5),          # across two int32 folds
])
def test_fused_int8_launch_of_a_layers_statistics(ops, dev, widths, heads, T, kind, planes, i8_route):
    """mdg_cov_accum_i8_multi: sigma_mlp, sigma_x and the per-head sigma_q / sigma_k (head_dim 128: diagonal tiles only) of one
    batch in ONE persistent launch over a shared tile schedule, one route for all.  Every statistic against the v_mfma_f64
    kernel entry-wise (within its own bound), a second call doubling the result, and the device route counters advancing by the number of
    statistics."""
    ops.i8_route_counts(dev, reset=True)
    items, refs, route, info = _fused_case(ops, dev, widths, heads, T, kind, 900 + T % 97)
    assert route == planes or (planes is None and route in (5, 6))
    planes = route
    fam = ["silu_gated" if (i == 0 and kind == "silu_gated") else "gaussian" for i in range(len(items))]
    for (S, _, _), R, i_, f_ in zip(items, refs, info, fam):
        _check_against(S, R, i_["bound"], f_)
    ops.cov_accum_i8_multi(items, report=False)
    for (S, _, _), R, i_, f_ in zip(items, refs, info, fam):
        _check_against(S, 2 * R, i_["bound"], f_)
    counts = ops.i8_route_counts(dev, reset=True)
    assert counts["i8_5"] + counts["i8_6"] + counts["fallback_f64"] == 2 * len(items) and counts["fallback_f64"] == 0
    if i8_route == "truncated":     # one route for the launch: the deepest any statistic asks for
        assert counts[{5: "i8_5", 6: "i8_6"}[planes]] == 2 * len(items) and counts["exact"] == 0
    else:                           # the exact route serves both classes: every statistic is booked under its own
        assert counts["exact"] == 2 * len(items) and all(i_["exact"] for i_ in info), (counts, info)
        # ... and the device picked the remainder implementation for the launch from the list lengths: Gaussian columns list ~1e-4 of
        # their elements (the tile kernel: up to 32 per column on average), a SiLU-gated statistic 0.5 % (the wide kernels, for every
        # statistic of its launch)
        # (the 12 000-token launch with one SiLU-gated statistic lists 33 elements per column on average, the threshold itself: either)
        assert all(i_["remainder"] == ("wide" if kind == "silu_gated" else "tiles") or (kind == "silu_gated" and T < 32768) for i_ in info), info
    assert counts["fp64_columns"] == 0 or T < 10240        # (short calls: the cross-term threshold is tighter, columns may leave)


def test_fused_int8_launch_lets_columns_and_statistics_leave_alone(ops, dev):
    """One column dominated by a single massive activation leaves the int8 path -- that COLUMN, through the fp64 column kernel;
    its statistic and the others of the launch stay on their digit planes.  A statistic whose every column is beyond six
    planes goes to the fp64 kernel alone (its result then equals ops.cov_accum's bit for bit)."""
    T = 12000
    Xa = gaussian(dev, T, 2048, 5)
    Xb = gaussian(dev, T, 2048, 6).clone()
    Xb[:, 17] = (Xb[:, 17].float() * 1e-6).to(torch.bfloat16)
    Xb[5, 17] = 300.0
    Xq = gaussian(dev, T, 4 * 128, 7).clone()
    Xq[:, 2 * 128 + 5] = (Xq[:, 2 * 128 + 5].float() * 1e-5).to(torch.bfloat16)      # head 2, feature 5
    Xq[100, 2 * 128 + 5] = -77.0
    items = [(torch.zeros(2048, 2048, dtype=F64, device=dev), Xa, 1), (torch.zeros(2048, 2048, dtype=F64, device=dev), Xb, 1),
             (torch.zeros(4, 128, 128, dtype=F64, device=dev), Xq, 4)]
    ops.i8_route_counts(dev, reset=True)
    info = []
    assert ops.cov_accum_i8_multi(items, report=True, route_info=info) == 5
    assert [i["columns"] for i in info] == [[], [17], [2 * 128 + 5]] and all(i["planes"] == 5 for i in info)
    assert ops.i8_route_counts(dev, reset=True) == {"i8_5": 3, "i8_6": 0, "fallback_f64": 0, "fp64_columns": 2, "exact": 0}
    refs = []
    for S, X, nh in items:
        R = torch.zeros_like(S)
        ops.cov_accum(R, X, n_heads=nh)
        refs.append(R)
    for (S, _, _), R, i_ in zip(items, refs, info):
        _check_against(S, R, i_["bound"], "outliers" if i_["columns"] else "gaussian")
    assert not torch.equal(items[0][0], refs[0])           # (an int8 result: equal within its bound, not to the bit)
    # a statistic with nothing the int8 path can certify goes to the fp64 kernel, alone
    Xh = (torch.randn(T, 2048, device=dev) ** 5 * torch.randn(T, 2048, device=dev) ** 3).to(torch.bfloat16)
    Rh = torch.zeros(2048, 2048, dtype=F64, device=dev)
    ops.cov_accum(Rh, Xh)
    mixed = [(torch.zeros(2048, 2048, dtype=F64, device=dev), Xa, 1), (torch.zeros(2048, 2048, dtype=F64, device=dev), Xh, 1)]
    assert ops.cov_accum_i8_multi(mixed, report=True) == 5
    assert ops.i8_route_counts(dev, reset=True) == {"i8_5": 1, "i8_6": 0, "fallback_f64": 1, "fp64_columns": 0, "exact": 0}
    assert torch.equal(mixed[1][0], Rh)
    _check_against(mixed[0][0], refs[0])
    both = [(torch.zeros(2048, 2048, dtype=F64, device=dev), Xh, 1), (torch.zeros(2048, 2048, dtype=F64, device=dev), Xh, 1)]
    assert ops.cov_accum_i8_multi(both, report=True) == 0
    assert torch.equal(both[0][0], Rh) and torch.equal(both[1][0], Rh)


def massive(X, cols, spikes=3, gap=12, seed=0):
    """BOS-like columns: the bulk of column c pushed `gap`+ binades under `spikes` massive activations."""
    g = torch.Generator(device=X.device).manual_seed(seed)
    X = X.clone()
    for i, c in enumerate(cols):
        top = X[:, c].float().abs().max()
        X[:, c] = (X[:, c].float() * 2.0 ** -(gap + i)).to(torch.bfloat16)
        rows = torch.randperm(X.shape[0], device=X.device, generator=g)[:spikes]
        X[rows, c] = (top * (1.0 + torch.rand(spikes, device=X.device, generator=g))).to(torch.bfloat16)
    return X


@pytest.mark.parametrize("n,kind,planes", [(4096, "gaussian", 5), (14336, "gaussian", 5), (14336, "silu_gated", 6)])
def test_massive_activation_columns_leave_alone_at_product_widths(ops, dev, n, kind, planes):
    """four BOS-like columns (bulk 12-15 binades under 3 spikes) at the widths of sigma_x and sigma_mlp.  The
    launch stays on its digit planes, exactly those four columns go to the fp64 column kernel, every entry equals the fp64
    kernel's within the call's bound (and the empirical 1e-12 of this measured family), and the call costs less than 5 % more than
    the clean one."""
    cols = [5, 129, n // 2 + 77, n - 1]
    X0 = (gaussian if kind == "gaussian" else silu_gated)(dev, T_BATCH, n, 300 + n)
    X = massive(X0, cols)
    S8 = torch.zeros(n, n, dtype=F64, device=dev)
    S64 = torch.zeros_like(S8)
    info = {}
    assert ops.cov_accum_i8(S8, X, route_info=info) == planes
    assert sorted(info["columns"]) == cols and info["bound"] <= 1.1e-11, info
    ops.cov_accum(S64, X)
    check_i8_error(entrywise_err(S8, S64), info["bound"], family="outliers", ctx=info)
    # the clean batch takes the same planes, no columns
    info0 = {}
    assert ops.cov_accum_i8(S8, X0, route_info=info0) == planes and info0["columns"] == []

    def once(Xin):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.cov_accum_i8(S8, Xin, report=False)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)
    S8.zero_()
    once(X0), once(X)                                    # warm-up of both
    t_clean = t_massive = 1e9
    for _ in range(5):                                   # alternating, best of five each: clock drift hits both alike
        t_clean, t_massive = min(t_clean, once(X0)), min(t_massive, once(X))
    print(f"n={n} {kind}: clean {t_clean:.2f} ms, with 4 massive columns {t_massive:.2f} ms (+{100 * (t_massive / t_clean - 1):.1f} %)")
    assert t_massive < 1.05 * t_clean + 0.4, (t_clean, t_massive)   # (+0.4 ms: at 4096 features the whole call is ~3 ms)


def test_device_route_equals_the_host_model_at_sigma_x_width(ops, dev):
    """The device's decision (planes, columns in greedy order, both parts of the bound) against tests/i8_model.py on the same data,
    4096 features x 12288 tokens, clean and with massive columns; measured error <= computed bound."""
    from tests import i8_model as M
    for kind, cols in (("gaussian", []), ("gaussian", [7, 2000]), ("silu_gated", [4095])):
        X = (gaussian if kind == "gaussian" else silu_gated)(dev, 12288, 4096, 17)
        X = massive(X, cols, gap=11) if cols else X
        S8 = torch.zeros(4096, 4096, dtype=F64, device=dev)
        S64 = torch.zeros_like(S8)
        info = {}
        ops.cov_accum_i8(S8, X, route_info=info)
        want = M.route_of(X.cpu(), offer_exact=ops.I8_EXACT)
        assert (info["planes"], info["columns"], info["exact"]) == (want["planes"], want["columns"], want["exact"]), \
            (kind, cols, info, want["planes"], want["columns"], want["exact"])
        assert abs(info["sq"] - want["sq"]) <= 1e-9 * want["sq"] and abs(info["x"] - want["x"]) <= 1e-9 * want["x"]
        ops.cov_accum(S64, X)
        check_i8_error(entrywise_err(S8, S64), info["bound"], family="outliers" if cols else kind, ctx=(kind, cols, info))


def test_cov_accum_multi_routes_a_llama_layer(ops, dev, monkeypatch):
    """ops.cov_accum_multi in "i8" mode (what the adapter hooks call): sigma_mlp in a launch and on a route of its own (six
    planes on SiLU-gated data), sigma_x + sigma_q + sigma_k together in one cov_accum_i8_multi launch (five planes); with
    MODEGPT_I8_FUSE off the per-head statistics go through the fp64 kernel instead -- same results within the route's bound."""
    T, f, d, nh, nkv = 12288, 4096, 2048, 8, 2
    H, X, Q, K = silu_gated(dev, T, f, 1), gaussian(dev, T, d, 2), gaussian(dev, T, nh * 128, 3), gaussian(dev, T, nkv * 128, 4)

    def run(fuse):
        monkeypatch.setattr(ops, "I8_FUSE", fuse)
        S = [torch.zeros(f, f, dtype=F64, device=dev), torch.zeros(d, d, dtype=F64, device=dev),
             torch.zeros(nh, 128, 128, dtype=F64, device=dev), torch.zeros(nkv, 128, 128, dtype=F64, device=dev)]
        ops.i8_route_counts(dev, reset=True)
        ops.cov_accum_multi([(S[0], H, 1), (S[1], X, 1), (S[2], Q, nh), (S[3], K, nkv)], mode="i8")
        return S, ops.i8_route_counts(dev, reset=True)

    fused, c1 = run(True)
    apart, c0 = run(False)
    # sigma_mlp alone (six-plane class: the exact route, the faster product there); x, q, k share a truncated five-plane launch
    assert c1 == {"i8_5": 3, "i8_6": 1, "fallback_f64": 0, "fp64_columns": 0, "exact": 1}
    assert c0 == {"i8_5": 1, "i8_6": 1, "fallback_f64": 0, "fp64_columns": 0, "exact": 1}   # separate launches: sigma_mlp, sigma_x; heads fp64
    for i, (a, b) in enumerate(zip(fused, apart)):
        _check_against(a, b, None, "silu_gated" if i == 0 else "gaussian")


def test_wide_remainder_kernels_at_their_edges(ops, dev, monkeypatch):
    """The dense-list implementation of the exact route's remainder (i8_lo_wide_kernel: one wave per column and 512 partner columns)
    where its indexing is not the plain case: a statistic whose width is not a multiple of 512 (the last partner block holds 128
    columns), columns the route hands to the fp64 column kernel in the middle of SiLU-gated data (rows / columns the remainder must
    leave alone), and per-head statistics that are dense themselves (the partner block is the head).  Every statistic against the
    v_mfma_f64 kernel within its own bound; the calls say the wide kernels ran."""
    monkeypatch.setattr(ops, "I8_EXACT", True)
    T = 16384                                               # ~80 listed elements per SiLU-gated column: the wide kernels
    # (a) 640 = 512 + 128 columns
    X = silu_gated(dev, T, 640, 31)
    S, R, info = torch.zeros(640, 640, dtype=F64, device=dev), torch.zeros(640, 640, dtype=F64, device=dev), {}
    ops.cov_accum_i8(S, X, route_info=info)
    ops.cov_accum(R, X)
    assert info["exact"] and info["remainder"] == "wide", info
    check_i8_error(entrywise_err(S, R), info["bound"], family="silu_gated")
    # (b) massive-activation columns leave; the others stay exact
    Xm = massive(silu_gated(dev, T, 2048, 32), [5, 700, 2047], gap=14)
    S, R, info = torch.zeros(2048, 2048, dtype=F64, device=dev), torch.zeros(2048, 2048, dtype=F64, device=dev), {}
    ops.cov_accum_i8(S, Xm, route_info=info)
    ops.cov_accum(R, Xm)
    assert info["exact"] and info["remainder"] == "wide" and sorted(info["columns"]) == [5, 700, 2047], info
    check_i8_error(entrywise_err(S, R), info["bound"], family="outliers")
    # (c) a launch whose per-head statistics are SiLU-gated too
    items, refs = [(torch.zeros(2048, 2048, dtype=F64, device=dev), silu_gated(dev, T, 2048, 33), 1)], []
    for j, nh in enumerate((6, 2)):
        items.append((torch.zeros(nh, 128, 128, dtype=F64, device=dev), silu_gated(dev, T, nh * 128, 34 + j), nh))
    for sigma, Xi, nh in items:
        Ri = torch.zeros_like(sigma)
        ops.cov_accum(Ri, Xi, n_heads=nh)
        refs.append(Ri)
    infos = []
    ops.cov_accum_i8_multi(items, report=True, route_info=infos)
    assert all(i_["exact"] and i_["remainder"] == "wide" for i_ in infos), infos
    for (Si, _, _), Ri, i_ in zip(items, refs, infos):
        _check_against(Si, Ri, i_["bound"], "silu_gated")
