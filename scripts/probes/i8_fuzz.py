"""Exploratory fuzz of mdg_cov_accum_i8 against the fp64 kernel (GPU): random shapes, sparsity, tails, outliers, exponent
range.  Prints every case whose entry-wise error over sqrt(sigma_ii sigma_jj) exceeds 1e-12 OR the bound the call itself computed
(guaranteed part: must never happen; the 1e-12 is the measured-typical claim), the route histogram, the largest measured / bound
ratio and how many columns went to the fp64 column kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops
dev = torch.device("cuda:0"); F64 = torch.float64
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
worst, routes, bad, worst_ratio, cols_out, worst_bound = 0.0, {}, 0, 0.0, 0, 0.0
exact_calls = wide_calls = 0
TMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 6000          # python i8_fuzz.py <seed> <cases> [max tokens] [max width / 128]
NMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 4
TOL = float(os.environ.get("MODEGPT_I8_TOLERANCE", "1"))        # (the route's tolerance dial scales the measured-typical limit with it)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    T = int(torch.randint(1, TMAX, (1,), generator=g)); n = 128 * int(torch.randint(1, NMAX + 1, (1,), generator=g))
    kind = int(torch.randint(0, 9, (1,), generator=g))
    z = torch.randn(T, n, generator=g)
    if kind == 1: z = z * (torch.rand(T, n, generator=g) < torch.rand(1, generator=g) * 0.5 + 0.01)
    elif kind == 2: z = z.abs()
    elif kind == 3: z = torch.nn.functional.silu(z) * torch.randn(T, n, generator=g)
    elif kind == 4: z = z ** 3
    elif kind == 5: z[torch.randint(0, T, (3,), generator=g), torch.randint(0, n, (3,), generator=g)] *= 10.0 ** float(torch.randint(1, 6, (1,), generator=g))
    elif kind == 6: z = torch.distributions.StudentT(3.0).sample((T, n))
    elif kind == 7: z = z * torch.exp(2 * torch.randn(T, 1, generator=g))          # token-wise scale mixture
    elif kind == 8: z = torch.round(z * 4) / 4                                       # few distinct values, many exact zeros
    expo = torch.randint(-100, 101, (n,), generator=g).double()
    X = (z.double() * torch.pow(torch.tensor(2.0, dtype=F64), expo)).to(torch.bfloat16).to(dev)
    S8 = torch.zeros(n, n, dtype=F64, device=dev); S64 = torch.zeros_like(S8)
    info = {}
    r = ops.cov_accum_i8(S8, X, route_info=info); ops.cov_accum(S64, X)
    cols_out += len(info["columns"])
    routes[r] = routes.get(r, 0) + 1
    d = torch.sqrt(torch.diag(S64)); d = torch.where(d > 0, d, torch.ones_like(d))
    low = torch.tril(torch.ones(n, n, dtype=torch.bool, device=dev))
    err = (((S8 - S64).abs() / (d[:, None] * d[None]))[low]).max().item()
    worst = max(worst, err)
    if r:
        if not info.get("exact"):
            worst_ratio = max(worst_ratio, err / max(info["bound"], 1e-300) if err > 4e-16 else 0.0)
        worst_bound = max(worst_bound, info["bound"])
    # (a call on the exact route is closer to the true sum than the fp64 kernel it is compared with, whose own rounding is up to
    #  ~2e-13 of sqrt(sigma_ii sigma_jj) at these sizes: tests/i8_limits.REFERENCE_ROUNDING)
    exact_calls += bool(info.get("exact"))
    wide_calls += info.get("remainder") == "wide"
    if not err < 1e-12 * TOL or (r and err > info["bound"] + (3e-13 if info.get("exact") else 4e-16)):
        bad += 1
        print(f"VIOLATION trial {trial}: T={T} n={n} kind={kind} route={r} err={err:.2e} bound={info['bound']:.2e} columns={info['columns']}")
print(f"routes {routes}; worst error {worst:.2e}; largest bound {worst_bound:.2e}; largest measured / bound {worst_ratio:.3f}; "
      f"columns sent to the fp64 column kernel {cols_out}; calls on the exact route {exact_calls} (remainder on the wide kernels: {wide_calls}); violations {bad}")
