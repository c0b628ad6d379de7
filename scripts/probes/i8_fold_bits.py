"""SHA-256 of the int8-route covariance of fixed inputs (two accumulating calls, three shapes): run under two builds of the library
(e.g. -DMDG_I8_FOLD_ATOMIC=0 / 1) to check that they agree bit for bit."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops

dev = torch.device("cuda:0")
h = hashlib.sha256()
for n, T, gated in ((4096, 5000, False), (2944, 70000, True), (8192, 3000, False)):
    g = torch.Generator(device=dev).manual_seed(n)
    X = torch.randn(T, n, device=dev, generator=g)
    if gated:
        X = torch.nn.functional.silu(X) * torch.randn(T, n, device=dev, generator=g)
    X = (X * torch.exp(torch.empty(n, device=dev).uniform_(-3.0, 0.7, generator=g))).to(torch.bfloat16)
    S = torch.zeros(n, n, dtype=torch.float64, device=dev)
    r1 = ops.cov_accum_i8(S, X)
    r2 = ops.cov_accum_i8(S, X[: T // 2])
    low = torch.tril(S)
    h.update(low.cpu().numpy().tobytes())
    print(n, T, "routes", r1, r2, "sum", float(low.sum()))
print("sha256", h.hexdigest())
