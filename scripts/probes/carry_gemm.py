"""The forward carry of the Nystrom substitution, X[Jend:] -= L[Jend:, J] W_J (7987 x 4096 x 2048 at r = 10035), measured against
its neighbours in shape: what makes it slower than the backward carry (53 against 67 TF in the trace of mdg_potrs_lower)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops
dev = torch.device("cuda:0"); F64 = torch.float64
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
g = torch.Generator(device=dev).manual_seed(0)
for (M, K, ld, nrhs, beta, what) in ((7987, 2048, 10048, 4096, 1.0, "as in potrs (ragged M, pitch 10048)"), (8192, 2048, 10048, 4096, 1.0, "M = 8192"),
                                     (7987, 2048, 14336, 4096, 1.0, "pitch 14336"), (7987, 2048, 10048, 4096, 0.0, "beta = 0"),
                                     (7987, 1024, 10048, 4096, 1.0, "K = 1024"), (7936, 2048, 10048, 4096, 1.0, "M = 7936 (62 full tile rows)")):
    L = torch.randn(M, ld, device=dev, generator=g, dtype=F64)[:, :K]
    W = torch.randn(K, nrhs, device=dev, generator=g, dtype=F64)
    X = torch.zeros(M, nrhs, device=dev, dtype=F64)
    t = timeit(lambda: ops.gemm(L, W, X, alpha=-1.0, beta=beta))
    print(f"carry {M} x {nrhs} x {K}, {what}: {t*1e6:.0f} us  {2.0*M*nrhs*K/t/1e12:.1f} TF")
