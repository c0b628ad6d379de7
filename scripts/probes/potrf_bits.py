"""SHA-256 of the blocked Cholesky's outputs (factor + inverted diagonal blocks) on fixed inputs: run under two builds of the library
(e.g. -DMDG_CHOL_TILE_KERNELS=0 / 1) to check that they agree bit for bit.  Also prints the time of one factorisation."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops

dev = torch.device("cuda:0")
h = hashlib.sha256()
for n in (640, 1025, 2500, 6016, 10035):
    g = torch.Generator(device=dev).manual_seed(n)
    X = torch.randn(2 * n, n, device=dev, generator=g, dtype=torch.float64)
    A = X.T @ X / (2 * n)
    A.diagonal().add_(1e-3)
    del X
    L = A.clone()
    inv = ops.potrf_lower(L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L2 = A.clone()
    ops.potrf_lower(L2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h.update(torch.tril(L).cpu().numpy().tobytes())
    h.update(inv[:-16].cpu().numpy().tobytes())
    print(n, "sum(L)", float(torch.tril(L).sum()), "%.2f ms (incl. the clone)" % (dt * 1e3))
print("sha256", h.hexdigest())
