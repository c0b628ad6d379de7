"""CPU simulation of the int8 digit-plane (Ozaki) covariance discussed in DESIGN.md section 7: accuracy of 5 planes with
the 15 plane pairs s + t <= 4 kept, against the fp64 product, on bench-like activations plus an outlier column."""
import torch, numpy as np
torch.manual_seed(0)
T, n, S = 8192, 96, 5
c = torch.exp(torch.empty(n).uniform_(np.log(0.05), np.log(2.0)))
X = (torch.randn(T, n) * c).to(torch.bfloat16)
X[::7, 3] *= 1e-6          # some tiny entries
X[5, 10] = 300.0           # an outlier
Xd = X.double()
ref = Xd.T @ Xd
# exact reference in higher precision via integer arithmetic (python ints through object arrays would be slow) -> use float128-ish: split sum
bits = X.view(torch.int16).to(torch.int32) & 0xFFFF
sign = torch.where((bits >> 15) & 1 == 1, -1, 1)
e = (bits >> 7) & 0xFF
m = bits & 0x7F
sig = torch.where(e > 0, m + 128, m) * sign          # |sig| <= 255
ee = torch.clamp(e, min=1)                           # effective exponent; value = sig * 2^(ee - 127 - 7)
E = ee.max(dim=0).values                             # per column
sh = (E[None, :] - ee).to(torch.int64)
W = 8 * S
N = torch.where(sh <= W - 10, sig.to(torch.int64) << torch.clamp(W - 10 - sh, min=0),
                torch.sign(sig).to(torch.int64) * ((sig.abs().to(torch.int64) + (1 << torch.clamp(sh - (W - 10) - 1, min=0, max=62))) >> torch.clamp(sh - (W - 10), min=0, max=63)))
digits = []
R = N.clone()
for s in range(S - 1):
    b = ((R + 128) & 0xFF) - 128          # balanced byte in [-128, 127]
    digits.append(b)
    R = (R - b) >> 8
digits.append(R)
digits = digits[::-1]                      # digits[0] most significant
assert all(int(d.abs().max()) <= 128 for d in digits), [int(d.abs().max()) for d in digits]
print("max |digit| per plane:", [int(d.abs().max()) for d in digits], " top plane range ok:", int(digits[0].abs().max()) <= 64)
# value = 2^(E - 127 - 7 - (W - 10)) * N ;  N = sum_s d_s 256^(S-1-s)
acc = torch.zeros(n, n, dtype=torch.float64)
for k in range(S):
    part = torch.zeros(n, n, dtype=torch.int64)
    for s in range(k + 1):
        t = k - s
        if s < S and t < S:
            part += digits[s].T @ digits[t]
    acc += part.double() * 2.0 ** (8 * (2 * (S - 1) - k))
scale = 2.0 ** (E.double() - 127 - 7 - (W - 10))
got = acc * scale[:, None] * scale[None, :]
err = (got - ref).abs().max() / ref.abs().max()
print("planes", S, "rel err vs fp64 matmul:", err.item())
d = torch.diag(ref)
print("diag rel err max:", ((torch.diag(got) - d).abs() / d).max().item())
rel = (got - ref).abs() / (torch.sqrt(torch.diag(ref))[:, None] * torch.sqrt(torch.diag(ref))[None, :])
i, j = divmod(int(rel.argmax()), n)
print("worst entry", i, j, rel[i, j].item(), "col max exps", int(E[i]), int(E[j]), "col scale", c[i].item(), c[j].item())
# exact integer check of the full (all pairs) expansion
full = torch.zeros(n, n, dtype=torch.float64)
for s in range(S):
    for t in range(S):
        full += (digits[s].T @ digits[t]).double() * 2.0 ** (8 * (2 * (S - 1) - s - t))
gotf = full * scale[:, None] * scale[None, :]
print("all 25 pairs: rel err", ((gotf - ref).abs().max() / ref.abs().max()).item())
