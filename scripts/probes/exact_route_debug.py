"""Where does the exact route's result differ from the truncated product / the fp64 kernel?  (debug probe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from modegpt_amd import ops
dev = torch.device("cuda:0"); F64 = torch.float64
def acts(gen, tokens, feat):
    z = torch.randn(tokens, feat, generator=gen); c = torch.exp(torch.empty(feat).uniform_(np.log(0.05), np.log(2.0), generator=gen)); return (z * c).to(torch.bfloat16)
def run(X, exact):
    ops.I8_EXACT = exact
    S = torch.zeros(X.shape[1], X.shape[1], dtype=F64, device=dev); info = {}
    ops.cov_accum_i8(S, X.to(dev), route_info=info)
    return torch.tril(S).cpu(), info
for T, n, r0, rows, c0, cols in ((6144, 256, 2048, 60, 64, 32), (6144, 256, 2048, 1, 64, 32), (6144, 256, 2048, 3, 64, 1), (6144, 256, 2048, 2, 64, 2),
                                 (6144, 256, 2048, 60, 64, 1), (6144, 256, 0, 60, 0, 32), (6144, 256, 2048, 33, 64, 32), (6144, 384, 2048, 60, 64, 32)):
    gen = torch.Generator().manual_seed(5)
    X = acts(gen, T, n)
    X[r0:r0 + rows, c0:c0 + cols] = (X[r0:r0 + rows, c0:c0 + cols].float() * 2.0 ** -20).to(torch.bfloat16)
    Se, ie = run(X, True); St, it = run(X, False)
    ref = torch.tril(X.double().T @ X.double())
    d = torch.sqrt(torch.diagonal(ref)); nrm = d[:, None] * d[None]
    ee, et = (Se - ref).abs() / nrm, (St - ref).abs() / nrm
    blk = torch.zeros(n, dtype=torch.bool); blk[c0:c0 + cols] = True
    inblk = blk[:, None] | blk[None, :]
    i, j = divmod(int(ee.argmax()), n)
    print(f"T={T} n={n} deep {rows}x{cols} at ({r0},{c0}): exact={ie['exact']} planes={ie['planes']}  err exact {ee.max():.2e} (in block rows/cols {ee[inblk].max():.2e}, elsewhere {ee[~inblk].max():.2e}) at ({i},{j});"
          f" truncated {et.max():.2e}")
    if ee.max() > 1e-12:
        k = (ee > 1e-12)
        print("   rows with error:", sorted(set(k.nonzero()[:, 0].tolist()))[:20], " cols:", sorted(set(k.nonzero()[:, 1].tolist()))[:20], " count", int(k.sum()))
        # the missing amount against the true remainder contribution of the deep block
        Xlo = torch.zeros_like(X, dtype=F64); Xlo[r0:r0 + rows, c0:c0 + cols] = X[r0:r0 + rows, c0:c0 + cols].double()
        contrib = torch.tril(Xlo.T @ X.double() + X.double().T @ Xlo - Xlo.T @ Xlo)
        miss = Se - ref
        print("   (exact - ref) / (block's whole contribution), at the worst entry:", (miss[i, j] / contrib[i, j]).item() if contrib[i, j] != 0 else None)
