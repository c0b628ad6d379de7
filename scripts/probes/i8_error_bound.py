"""CPU model of the error BOUND of the int8 digit-plane covariance (cov_i8.hip), next to its measured error.

Digits: x_ti = 2^(E_i - 172) N_ti, N_ti = sum_s d_s(t, i) 256^(5 - s) exactly for elements within 38 binades of their column
maximum (deeper ones are rounded to an integer: remainder term rho).  The P-plane product keeps the plane pairs s + t < P, so

    err_ij = 2^(E_i + E_j - 344) sum_{s + t >= P} 256^(10 - s - t) <d_s(., i), d_t(., j)>      (+ the rounding remainder)

and by Cauchy-Schwarz, with alpha_s(i) = 256^(5 - s) ||d_s(., i)|| / ||N_i||  (plane energies relative to the column norm),

    |err_ij| / sqrt(sigma_ii sigma_jj)  <=  sum_{s + t >= P} alpha_s(i) alpha_t(j)  + rho_i + rho_j + rho_i rho_j   =: bound_P(i, j)
                                        <=  sum_{s + t >= P} A_s A_t + ...,   A_s = max_j alpha_s(j)                =: beta_P

(sigma_ii of THIS batch; summed over batches the same inequality holds against the accumulated diagonal, Cauchy-Schwarz again).
The bound is attained by the squared terms on the diagonal (s = t, i = j: sum_t d_s^2 is what it is) and by duplicated / strongly
correlated columns; for independent columns the cross sums grow like sqrt(tokens) instead of tokens and the error sits far below.

Prints, per distribution family: the old route statistic (deep share), the measured error of P = 5 / 6 and the bounds.
"""
import sys

import numpy as np
import torch

torch.manual_seed(0)
T, n, NP, TOP = 32768, 48, 6, 38
F64 = torch.float64


def parts(X):
    bits = X.view(torch.int16).to(torch.int32) & 0xFFFF
    sign = torch.where((bits >> 15) & 1 == 1, -1, 1)
    e = (bits >> 7) & 0xFF
    m = bits & 0x7F
    sig = (torch.where(e > 0, m + 128, m) * sign).to(torch.int64)
    ee = torch.clamp(e, min=1).to(torch.int64)
    return sig, ee


def digits_of(X, E=None):
    sig, ee = parts(X)
    if E is None:
        E = torch.where(sig != 0, ee, torch.ones_like(ee)).max(dim=0).values
    sh = E[None, :] - ee
    up = sig << torch.clamp(TOP - sh, min=0)
    dn = torch.clamp(sh - TOP, min=1, max=62)
    mag = torch.where(dn > 9, torch.zeros_like(sig), (sig.abs() + (1 << (dn - 1))) >> dn)
    N = torch.where(sh <= TOP, up, torch.sign(sig) * mag)
    rounded = ((sh > TOP) & (sig != 0)).sum(0)
    ds, R = [], N.clone()
    for _ in range(NP - 1):
        b = ((R + 128) & 0xFF) - 128
        ds.append(b)
        R = (R - b) >> 8
    ds.append(R)
    return ds[::-1], E, N, rounded          # ds[0] most significant


def product(ds, E, P, keep=None):
    cls = [torch.zeros(ds[0].shape[1], ds[0].shape[1], dtype=torch.int64) for _ in range(11)]
    for s in range(NP):
        for t in range(NP):
            if (s + t < P) if keep is None else keep(s, t):
                cls[s + t] += ds[s].T @ ds[t]
    acc = sum(c.double() * 2.0 ** (80 - 8 * k) for k, c in enumerate(cls))
    sc = torch.pow(torch.tensor(2.0, dtype=F64), (E - 172).double())
    return acc * sc[:, None] * sc[None, :]


def alphas(ds, N, rounded):
    norm = torch.sqrt((N.double() ** 2).sum(0)).clamp(min=1.0)
    a = torch.stack([torch.sqrt((d.double() ** 2).sum(0)) * 256.0 ** (NP - 1 - s) / norm for s, d in enumerate(ds)])   # [6, n]
    rho = 0.5 * torch.sqrt(rounded.double()) / norm
    return a, rho


def bound(a, rho, P, entrywise=False):
    if entrywise:
        b = sum(a[s][:, None] * a[t][None, :] for s in range(NP) for t in range(NP) if s + t >= P)
        return b + rho[:, None] + rho[None, :] + rho[:, None] * rho[None, :]
    A = a.max(dim=1).values
    r = rho.max()
    return (sum(A[s] * A[t] for s in range(NP) for t in range(NP) if s + t >= P) + 2 * r + r * r).item()


def dist(kind, n=n):
    g, u, w = torch.randn(T, n), torch.randn(T, n), torch.randn(T, n)
    c = torch.exp(torch.empty(n).uniform_(np.log(0.05), np.log(2.0)))
    x = {"gaussian": g, "silu_gated": torch.nn.functional.silu(g) * u, "gelu_gated": torch.nn.functional.gelu(g) * u,
         "laplace": torch.sign(g) * torch.log(torch.rand(T, n)), "student_t4": g / torch.sqrt((torch.randn(4, T, n) ** 2).mean(0)),
         "prod2": g * u, "prod3": g * u * w, "relu": torch.relu(g), "silu2": torch.nn.functional.silu(g) * u * (0.3 + w.abs()),
         "g^2u": g * g * u, "laplace^1.5": torch.sign(g) * torch.log(torch.rand(T, n)).abs() ** 1.5, "g^3": g ** 3}[kind]
    return (x * c).to(torch.bfloat16)


def massive(kind, spikes=3, gap=12):
    """columns 0..3: a bulk `gap` binades under `spikes` massive activations (BOS-like); the rest as `kind`."""
    X = dist(kind).float()
    for j in range(4):
        rows = torch.randperm(T)[:spikes]
        X[rows, j] = X[:, j].abs().max() * 2.0 ** (gap + j)
    return X.to(torch.bfloat16)


def report(name, X):
    ds, E, N, rounded = digits_of(X)
    ref = X.double().T @ X.double()
    d = torch.sqrt(torch.diag(ref))
    sig, ee = parts(X)
    nz = sig != 0
    share = (((E[None, :] - ee) >= 10) & nz).sum(0).double() / nz.sum(0).clamp(min=1)
    a, rho = alphas(ds, N, rounded)
    crest = (X.double().abs().max(0).values / torch.sqrt((X.double() ** 2).mean(0))).max().item()
    out = [f"{name:22s} share {share.max().item():5.3f} crest {crest:6.1f}"]
    for P in (5, 6):
        got = product(ds, E, P)
        rel = (got - ref).abs() / (d[:, None] * d[None, :])
        e_all, e_diag = rel.max().item(), torch.diag(rel).max().item()
        be = bound(a, rho, P, entrywise=True)
        assert bool((rel <= be * (1 + 1e-9) + 2e-15).all()), (name, P, (rel / be).max().item())
        out.append(f"| P{P}: err {e_all:8.2e} (diag {e_diag:8.2e})  bound {bound(a, rho, P):8.2e}  tightest ratio {(rel / be.clamp(min=1e-300)).max().item():6.3f}")
    print(" ".join(out))
    return a


if __name__ == "__main__":
    kinds = ("gaussian", "relu", "silu_gated", "gelu_gated", "laplace", "student_t4", "prod2", "silu2", "g^2u", "laplace^1.5", "prod3", "g^3")
    for k in kinds:
        a = report(k, dist(k))
        if "-v" in sys.argv:
            print("      A_s =", " ".join(f"2^{np.log2(max(v, 1e-300)):6.1f}" for v in a.max(dim=1).values.tolist()))
    for k in ("gaussian", "silu_gated"):
        for gap in (10, 15):
            a = report(f"{k}+massive(gap {gap})", massive(k, gap=gap))
            if "-v" in sys.argv:
                print("      A_s =", " ".join(f"2^{np.log2(max(v, 1e-300)):6.1f}" for v in a.max(dim=1).values.tolist()))
