"""CPU calibration of the depth statistic of csrc/cov_i8.hip: for several activation distributions, the error of the 5- and
6-plane int8 digit-plane covariance against the exact (fp64) product, entry-wise over sqrt(sigma_ii sigma_jj), next to the
largest per-column share of elements more than 10 binades below the column maximum."""
import torch, numpy as np
torch.manual_seed(0)
def split_err(X, NP, keep_classes):
    Xd = X.double(); ref = Xd.T @ Xd
    bits = X.view(torch.int16).to(torch.int32) & 0xFFFF
    sign = torch.where((bits >> 15) & 1 == 1, -1, 1)
    e = (bits >> 7) & 0xFF; m = bits & 0x7F
    sig = torch.where(e > 0, m + 128, m) * sign
    ee = torch.clamp(e, min=1)
    E = torch.where(sig != 0, ee, torch.ones_like(ee)).max(dim=0).values
    sh = (E[None] - ee).to(torch.int64)
    W = 8 * NP - 10
    N = torch.where(sh <= W, sig.to(torch.int64) << torch.clamp(W - sh, min=0), torch.zeros_like(sh))
    digs = []; R = N.clone()
    for s in range(NP - 1):
        b = ((R + 128) & 0xFF) - 128; digs.append(b); R = (R - b) >> 8
    digs.append(R); digs = digs[::-1]
    n = X.shape[1]; acc = torch.zeros(n, n, dtype=torch.float64)
    for k in range(keep_classes):
        part = torch.zeros(n, n, dtype=torch.float64)
        for s in range(min(k, NP - 1) + 1):
            t = k - s
            if t < NP: part += (digs[s].double().T @ digs[t].double())
        acc += part * 2.0 ** (8 * (2 * (NP - 1) - k))
    scale = 2.0 ** (E.double() - 134 - W)
    got = acc * scale[:, None] * scale[None]
    d = torch.sqrt(torch.diag(ref))
    err = ((got - ref).abs() / (d[:, None] * d[None])).max().item()
    nz = sig != 0
    deep = ((sh >= 10) & nz).double().mean(0).max().item()
    near = ((sh <= 5) & nz).double().mean(0).min().item()
    return err, deep, near
T, n = 8192, 64
g = torch.randn(T, n); u = torch.randn(T, n)
data = {"gauss": torch.randn(T, n), "silu(g)*u": torch.nn.functional.silu(g) * u, "gauss^3": torch.randn(T, n) ** 3,
        "gauss^5": torch.randn(T, n) ** 5, "laplace": torch.distributions.Laplace(0, 1).sample((T, n)), "relu(g)": torch.relu(g),
        "lognormal*sign": torch.exp(2 * torch.randn(T, n)) * torch.sign(torch.randn(T, n))}
for name, X in data.items():
    X = X.to(torch.bfloat16)
    e5, deep, near = split_err(X, 5, 5)
    e6, _, _ = split_err(X, 6, 6)
    e65, _, _ = split_err(X, 6, 5)
    print(f"{name:16s} err NP5={e5:.1e}  NP6(top5 planes, 5 classes)={e65:.1e}  NP6={e6:.1e}   max deep frac(sh>=10)={deep:.3f}  min near frac(sh<=5)={near:.3f}")
print("---- larger T, more shapes")
T, n = 32768, 48
g = torch.randn(T, n); u = torch.randn(T, n)
st = torch.distributions.StudentT(4.0).sample((T, n))
data = {"silu(g)*u T=32768": torch.nn.functional.silu(g) * u, "g*u": g * u, "g*|g|": g * g.abs(), "studentT4": st,
        "silu(g)*u * scale": torch.nn.functional.silu(g) * u * torch.exp(torch.randn(n)), "gelu(g)*u": torch.nn.functional.gelu(g) * u,
        "g^3": g ** 3, "g*u*v": g * u * torch.randn(T, n)}
for name, X in data.items():
    X = X.to(torch.bfloat16)
    e6, deep, near = split_err(X, 6, 6)
    e5, _, _ = split_err(X, 6, 5)
    print(f"{name:22s} err P5={e5:.1e} P6={e6:.1e}   max deep frac={deep:.3f}")
