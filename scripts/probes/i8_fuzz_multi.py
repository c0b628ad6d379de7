"""Exploratory fuzz of mdg_cov_accum_i8_multi against the fp64 kernel (GPU): 1-4 statistics per call -- full ones of random width,
per-head ones of random head count -- random token counts (1 .. 9000), nine distribution families, the whole bf16 exponent range
per column.  Prints every statistic whose entry-wise error over sqrt(sigma_ii sigma_jj) exceeds 1e-12, and the route histogram.
   usage: python3 i8_fuzz_multi.py [seed] [trials]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops
dev = torch.device("cuda:0"); F64 = torch.float64
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 11)


def sample(T, n):
    kind = int(torch.randint(0, 9, (1,), generator=g))
    z = torch.randn(T, n, generator=g)
    if kind == 1: z = z * (torch.rand(T, n, generator=g) < torch.rand(1, generator=g) * 0.5 + 0.01)
    elif kind == 2: z = z.abs()
    elif kind == 3: z = torch.nn.functional.silu(z) * torch.randn(T, n, generator=g)
    elif kind == 4: z = z ** 3
    elif kind == 5: z[torch.randint(0, T, (3,), generator=g), torch.randint(0, n, (3,), generator=g)] *= 10.0 ** float(torch.randint(1, 6, (1,), generator=g))
    elif kind == 6: z = torch.distributions.StudentT(3.0).sample((T, n))
    elif kind == 7: z = z * torch.exp(2 * torch.randn(T, 1, generator=g))
    elif kind == 8: z = torch.round(z * 4) / 4
    expo = torch.randint(-100, 101, (n,), generator=g).double()
    return (z.double() * torch.pow(torch.tensor(2.0, dtype=F64), expo)).to(torch.bfloat16).to(dev), kind


def err_of(S, R):
    d = torch.sqrt(torch.diagonal(R)); d = torch.where(d > 0, d, torch.ones_like(d))
    low = torch.tril(torch.ones_like(R, dtype=torch.bool))
    return (((S - R).abs() / (d[:, None] * d[None]))[low]).max().item()


worst, routes, bad, cols_out, worst_ratio = 0.0, {}, 0, 0, 0.0
exact_stats = wide_stats = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    T = int(torch.randint(1, 9000, (1,), generator=g))
    count = int(torch.randint(1, 5, (1,), generator=g))
    items, kinds = [], []
    for i in range(count):
        if i > 0 and int(torch.randint(0, 2, (1,), generator=g)):      # per-head statistic (never first: the launch needs a schedule anyway)
            nh = int(torch.randint(2, 12, (1,), generator=g))
            X, k = sample(T, nh * 128)
            items.append((torch.zeros(nh, 128, 128, dtype=F64, device=dev), X, nh))
        else:
            n = 128 * int(torch.randint(1, 20, (1,), generator=g))
            X, k = sample(T, n)
            items.append((torch.zeros(n, n, dtype=F64, device=dev), X, 1))
        kinds.append(k)
    infos = []
    r = ops.cov_accum_i8_multi(items, report=True, route_info=infos)
    cols_out += sum(len(i_["columns"]) for i_ in infos)
    routes[r] = routes.get(r, 0) + 1
    for (S, X, nh), k, info in zip(items, kinds, infos):
        R = torch.zeros_like(S)
        ops.cov_accum(R, X, n_heads=nh)
        e = err_of(S, R) if S.dim() == 2 else max(err_of(S[h], R[h]) for h in range(nh))
        worst = max(worst, e)
        # (the guaranteed part: must never happen; a statistic on the exact route is closer to the true sum than the fp64 kernel it is
        #  compared with, whose own rounding is up to ~2e-13 of sqrt(sigma_ii sigma_jj): tests/i8_limits.REFERENCE_ROUNDING)
        over_bound = info["planes"] != 0 and e > info["bound"] + (3e-13 if info.get("exact") else 4e-16)
        exact_stats += bool(info.get("exact")); wide_stats += info.get("remainder") == "wide"
        if info["planes"] and e > 4e-16:
            if not info.get("exact"):
                worst_ratio = max(worst_ratio, e / max(info["bound"], 1e-300))
        if not e < 1e-12 or over_bound:
            bad += 1
            print(f"VIOLATION trial {trial}: T={T} shape={tuple(S.shape)} kind={k} route={r} of {count} statistics err={e:.2e} "
                  f"bound={info['bound']:.2e} columns={info['columns']}")
print(f"routes {routes}; worst error {worst:.2e}; largest measured / bound {worst_ratio:.3f}; columns sent to the fp64 column kernel {cols_out}; "
      f"statistics on the exact route {exact_stats} (remainder on the wide kernels: {wide_stats}); violations {bad}")
