"""SHA-256 of the batched Jacobi eigensolver's outputs on fixed inputs (run under two builds, e.g. -DMDG_SYEVJ_LDS_V=0 / 1, to check
that they agree bit for bit) and its time for the 8 head-sized Grams of a Llama-3-8B layer."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops

dev = torch.device("cuda:0")
h = hashlib.sha256()
for b, n, rank in ((8, 128, 128), (5, 64, 64), (3, 128, 40), (2, 30, 30)):
    g = torch.Generator(device=dev).manual_seed(100 * n + b)
    X = torch.randn(b, 4 * n, rank, device=dev, generator=g, dtype=torch.float64) @ torch.randn(b, rank, n, device=dev, generator=g, dtype=torch.float64)
    A = X.transpose(1, 2) @ X / (4 * n)
    ev, V = ops.syevj(A)
    res = ((A @ V - V * ev[:, None, :]).abs().max() / A.abs().max()).item()
    h.update(ev.cpu().numpy().tobytes())
    h.update(V.cpu().numpy().tobytes())
    print(b, n, rank, "residual |A V - V L| / |A| = %.1e" % res, "sum(ev) %.12g" % float(ev.sum()))
print("sha256", h.hexdigest())
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(8, 4096, 128, device=dev, generator=g, dtype=torch.float64)
A = X.transpose(1, 2) @ X / 4096
ops.syevj(A)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ops.syevj(A)
e1.record()
torch.cuda.synchronize()
print("syevj 8 x 128 x 128: %.2f ms per call (incl. the input clone)" % (e0.elapsed_time(e1) / 3))
