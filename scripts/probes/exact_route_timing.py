"""Whole-call time of the int8 covariance at the sigma_mlp shape (32768 x 14336), exact route against the truncated product,
Gaussian and SiLU-gated columns.    python scripts/probes/exact_route_timing.py [n=14336] [tokens=32768] [gaussian|silu_gated] [exact|truncated]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops
dev = torch.device("cuda:0"); F64 = torch.float64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14336
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
def gaussian(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    c = torch.exp(torch.empty(n, device=dev).uniform_(math.log(0.05), math.log(2.0), generator=g))
    return (torch.randn(T, n, device=dev, generator=g) * c).to(torch.bfloat16)
def gated(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.nn.functional.silu(torch.randn(T, n, device=dev, generator=g)); a.mul_(torch.randn(T, n, device=dev, generator=g))
    return a.to(torch.bfloat16)
S = torch.zeros(n, n, dtype=F64, device=dev)
only = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "all" else None
routes = {"exact": (True,), "truncated": (False,)}.get(sys.argv[4] if len(sys.argv) > 4 else None, (True, False))
for name, make, seed in (("gaussian", gaussian, 1), ("silu_gated", gated, 2)):
    if only not in (None, name):
        continue
    X = make(seed)
    for exact in routes:
        ops.I8_EXACT = exact
        info, st = {}, {}
        cls = ops.cov_accum_i8(S, X, route_info=info, mfma_stats=st)
        for _ in range(2):
            ops.cov_accum_i8(S, X, report=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(); k1.record()
        reps = 6
        tot_k = 0.0
        e0.record()
        for _ in range(reps):
            ops.cov_accum_i8(S, X, report=False, events=(k0, k1))
        e1.record(); torch.cuda.synchronize()
        # (the product-launch events hold the LAST call's; time that one alone as well)
        print(f"{name:11s} exact={exact!s:5s} class {cls} exact_ran={info['exact']!s:5s} whole call {e0.elapsed_time(e1) / reps:7.3f} ms   "
              f"product launches (last call) {k0.elapsed_time(k1):7.3f} ms   executed/dense {st['executed'] / st['dense']:.3f} of the {st['planes_run']}-plane launch's pairs   bound {info['bound']:.2e}")
