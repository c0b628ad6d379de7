"""Host-side facts the int8 digit-plane covariance kernel (modegpt_amd/csrc/cov_i8.hip) relies on, checked by enumeration over
EVERY digit vector its split pass can emit (the arithmetic below restates i8_split_kernel's: a signed 8-bit significand
shifted to 48 bits, six balanced base-256 digits taken from the low end).  No GPU needed."""
import numpy as np

NP_, TOP_SHIFT, FLUSH_STEPS, KS = 6, 38, 2047, 32


def digit_vectors():
    out, values = [], []
    for sh in range(0, 50):
        for sig in range(-255, 256):
            if sig == 0:
                continue
            if sh <= TOP_SHIFT:
                N = sig << (TOP_SHIFT - sh)
            else:
                dn = sh - TOP_SHIFT
                mag = 0 if dn > 9 else (abs(sig) + (1 << (dn - 1))) >> dn
                N = -mag if sig < 0 else mag
            n0, d = N, [0] * NP_
            for s in range(NP_ - 1, 0, -1):
                b = ((N + 128) & 255) - 128
                d[s] = b
                N = (N - b) >> 8
            d[0] = N
            out.append(d)
            values.append(n0)
    return np.array(out, dtype=np.int64), np.array(values, dtype=np.int64)


def test_digits_are_an_exact_int8_representation_with_at_most_three_nonzero_planes():
    V, N = digit_vectors()
    assert V.min() >= -128 and V.max() <= 127                       # every digit fits int8
    assert np.abs(V[:, 0]).max() <= 64                              # the top digit has a spare bit
    weights = 256 ** np.arange(NP_ - 1, -1, -1, dtype=np.int64)
    assert np.array_equal(V @ weights, N)                           # error-free: the digits ARE the fixed-point value
    assert (V != 0).sum(1).max() <= 3                               # two full digits and a carry digit


def test_int32_classes_cannot_overflow_within_the_fold_interval():
    """|sum_{s+t=k} d_s(i) d_t(j)| <= 32768 for any two elements, so FLUSH_STEPS k-steps of 32 tokens stay below 2^31."""
    V = np.unique(digit_vectors()[0], axis=0)
    worst = 0
    for P in (5, 6):
        for k in range(P):
            W = np.zeros_like(V)
            for t in range(NP_):
                if 0 <= k - t < P and t < P:
                    W[:, t] = V[:, k - t]
            for lo in range(0, len(V), 4096):
                worst = max(worst, int(np.abs(W[lo:lo + 4096] @ V.T).max()))
    assert worst == 32768
    assert FLUSH_STEPS * KS * worst < 2 ** 31


def test_fold_interval_in_the_kernel_source_matches():
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "modegpt_amd", "csrc", "cov_i8.hip")).read()
    assert int(re.search(r"constexpr int FLUSH_STEPS = (\d+);", src).group(1)) == FLUSH_STEPS
    assert int(re.search(r"constexpr int TOP_SHIFT = 8 \* NP - (\d+)", src).group(1)) == 8 * NP_ - TOP_SHIFT
