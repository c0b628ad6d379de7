"""Exact int32 headroom of the digit-plane classes (csrc/cov_i8.hip).  Every bf16 element is a signed 8-bit significand at
some shift below its column's maximum, so its six balanced base-256 digits have at most two 'full' digits and a carry digit.
Enumerates ALL digit vectors the split pass can produce (sig -255..255, shift 0..47) and, per class k, the largest
|sum_{s+t=k} d_s(i) d_t(j)| over all pairs of elements -> the number of tokens a class can accumulate before 2^31 - 1.
"""
import numpy as np

NP_, TOP = 6, 38
vecs = set()
for sh in range(0, 49):
    for sig in range(-255, 256):
        if sig == 0:
            continue
        if sh <= TOP:
            N = sig << (TOP - sh)
        else:
            dn = sh - TOP
            mag = 0 if dn > 9 else (abs(sig) + (1 << (dn - 1))) >> dn
            N = -mag if sig < 0 else mag
        d = [0] * NP_
        for s in range(NP_ - 1, 0, -1):
            b = ((N + 128) & 255) - 128
            d[s] = b
            N = (N - b) >> 8
        d[0] = N
        assert -128 <= d[0] <= 127, (sig, sh, d)
        vecs.add(tuple(d))
V = np.array(sorted(vecs), dtype=np.int64)
print(len(V), "distinct digit vectors; |d_0| max", np.abs(V[:, 0]).max(), "; nonzero digits per element max", (V != 0).sum(1).max())
for P in (5, 6):
    worst = 0
    for k in range(P):
        W = np.zeros_like(V)                       # W[i][t] = d_{k-t}(i): class-k sum = W_i . V_j
        for t in range(NP_):
            if 0 <= k - t < P and t < P:
                W[:, t] = V[:, k - t]
        best = 0
        Vt = V[:, :].T.copy()
        for lo in range(0, len(V), 4096):
            best = max(best, int(np.abs(W[lo:lo + 4096] @ Vt).max()))
        worst = max(worst, best)
        print(f"P={P} class {k}: max |sum| per token = {best}")
    print(f"P={P}: a class holds {(2**31 - 1) // worst} tokens = {(2**31 - 1) // worst // 32} k-steps of 32 before int32 could overflow")
