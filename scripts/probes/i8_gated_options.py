"""VERDICT r2 item 3 -- can the six-plane launch of SiLU-gated activations (15.1 executed plane pairs, 35.9 ms) come down to ~12
pairs?  CPU model (tests/i8_model.py) of the three candidates on silu(g) * u columns, 32768 tokens:

  (a) element-level outlier split  X = X_c + R: the elements above a per-column quantile leave into a sparse fp64 residual, the
      clipped bulk gets a lower column maximum, hence shallower digits -- does FIVE planes then meet the bound?  Priced: the residual
      needs X^T R + R^T X - R^T R; organised per column that reads nnz(R) rows of X (28 KB each at 14336 features).
  (b) truncation depth per (32-column group I, group J) instead of per launch: how many tile pairs could run five planes?
  (c) "5 1/2 planes": five planes + the two dominant class-5 pairs (2,3), (3,2) from digits rounded to 4 bits inside class 4.

Result (round 3): none of them reaches the bound's thresholds (SQ <= 1e-12, X <= 1e-11) at a cost below what it saves; the table
this prints is in DESIGN.md section 7.  The reason is one fact: the crest factor of these columns comes from their BULK sitting
near zero (a product of two near-Gaussians), not from outliers -- the maximum of 32768 draws of an exponential-tailed variable is
only ~1.5x its 1e-3 quantile, so clipping buys 0.6 binades where 3 are needed, and every column group looks the same."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tests import i8_model as M

torch.manual_seed(0)
T, n = 32768, 64
g, u = torch.randn(T, n), torch.randn(T, n)
c = torch.exp(torch.empty(n).uniform_(np.log(0.05), np.log(2.0)))
X = (torch.nn.functional.silu(g) * u * c).to(torch.bfloat16)
Xd = X.double().numpy()
ref = Xd.T @ Xd
dd = np.sqrt(np.diag(ref))


def bound_and_error(Xb, P):
    d, E, N, rounded, nnz = M.digits(Xb)
    st = M.column_stats(d, rounded, nnz)
    a, rho = M.alphas(st)
    sq, x = M.terms(a.max(1), rho.max(), P)
    xb = Xb.double().numpy()
    r = xb.T @ xb
    d2 = np.sqrt(np.diag(r))
    err = (np.abs(M.product(d, E, P) - r) / (d2[:, None] * d2[None, :])).max()
    return sq, x, err, a


print(f"silu(g) * u, {T} tokens x {n} columns; thresholds: SQ <= {M.TAU_SQ:.0e}, X <= {M.TAU_X:.0e}")
for P in (5, 6):
    sq, x, err, a = bound_and_error(X, P)
    print(f"  as is, {P} planes: SQ {sq:.2e}  X {x:.2e}  measured {err:.2e}   A_s = 2^{np.round(np.log2(np.maximum(a.max(1), 1e-300)), 1)}")

print("(a) element-level outlier split, five planes on the clipped bulk:")
absx = np.abs(Xd)
for p in (1e-4, 1e-3, 1e-2, 3e-2):
    thr = np.quantile(absx, 1.0 - p, axis=0)
    keep = absx <= thr[None, :]
    Xc = torch.from_numpy(np.where(keep, Xd, 0.0)).to(torch.bfloat16)
    nnz = int((~keep).sum())
    sq, x, err, _ = bound_and_error(Xc, 5)
    binades = np.log2(absx.max(0) / np.abs(Xc.double().numpy()).max(0)).mean()
    # residual priced at the product's shape: 14336 features, 32768 tokens: per column nnz_col rows of X (28 KB) + one sigma column
    nnz_full = p * T * 14336
    print(f"  quantile 1 - {p:.0e}: column maximum drops {binades:.2f} binades; X_5 {x:.2e} (needs <= 1e-11), measured {err:.2e}; "
          f"residual at 14336 features: {nnz_full:.2e} nonzeros x 28 KB rows = {nnz_full * 28672 / 1e9:.0f} GB of row reads "
          f"= {nnz_full * 28672 / 4e12 * 1e3:.1f} ms at 4 TB/s")

print("(b) depth per (group I, group J) of 32 columns:")
d, E, N, rounded, nnz = M.digits(X)
st = M.column_stats(d, rounded, nnz)
a, rho = M.alphas(st)
G = n // 32
ok5 = 0
for I in range(G):
    for J in range(G):
        AI, AJ = a[:, I * 32:(I + 1) * 32].max(1), a[:, J * 32:(J + 1) * 32].max(1)
        x = sum(AI[s] * AJ[t] for s in range(6) for t in range(6) if s != t and s + t >= 5)
        ok5 += x <= M.TAU_X
print(f"  group pairs that meet X <= 1e-11 on five planes: {ok5} of {G * G}   (per-column X_5 of the single columns: "
      f"min {min(sum(a[s, j] * a[t, j] for s in range(6) for t in range(6) if s != t and s + t >= 5) for j in range(n)):.2e})")

print("(c) 5 1/2 planes (pairs (2,3), (3,2) from 4-bit digits inside class 4):")
A = a.max(1)
others = sum(A[s] * A[t] for s in range(6) for t in range(6) if s != t and s + t >= 5 and (s, t) not in ((2, 3), (3, 2)))
# d = 16 r + e, |e| <= 8: d2 d3 - 256 r2 r3 = 16 (r2 e3 + r3 e2) + e2 e3, bounded through ||e|| <= 8 sqrt(count) <= (8 / rms_d) ||d||
approx = 2 * A[2] * A[3] * (2 * 8 / 74.0 + (8 / 74.0) ** 2)
print(f"  X without the two pairs {others:.2e} + their approximation error <= {approx:.2e} = {others + approx:.2e} (needs <= 1e-11); "
      f"executed pairs 13 + 2 = 15 on the five-plane tile against 15.1 today: nothing saved even if it met the bound")
