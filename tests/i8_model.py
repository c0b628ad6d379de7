"""Host model of the int8 digit-plane covariance route (modegpt_amd/csrc/cov_i8.hip): the split pass's per-column integer
statistics, the Cauchy-Schwarz error bound built from them, and the greedy choice of (planes, columns sent to the fp64 column
kernel).  Test infrastructure: the CPU tests check the bound against the exact truncation error of the modelled product, the
GPU tests check the device's decisions against this model on the same inputs.

Digits (i8_split_vec_kernel): x_tj = 2^(E_j - 172) N_tj, N = sum_s d_s 256^(5 - s), six balanced base-256 digits; exact for
elements within 38 binades of the column maximum E_j, rounded to an integer below (`rounded` counts those).  The P-plane product
keeps the plane pairs s + t < P.  With

    alpha_s(j) = 256^(5 - s) ||d_s(., j)||_2 / ||N_j||_2          rho_j = sqrt(rounded_j) / (2 ||N_j||_2)

Cauchy-Schwarz over the tokens gives, entry-wise and for ANY input,

    |sigma_ij(P planes) - sigma_ij| / sqrt(sigma_ii sigma_jj)  <=  sum_{s + t >= P} alpha_s(i) alpha_t(j) + rho_i + rho_j + rho_i rho_j
                                                               <=  SQ_P + X_P
    SQ_P = sum_{2 s >= P} A_s^2 + 2 R + R^2      (attained on the diagonal: a sum of squares is what it is)
    X_P  = sum_{s != t, s + t >= P} A_s A_t      (attained only by columns whose digit sequences are proportional; for
                                                  independent columns the cross sums grow like sqrt(tokens), not tokens)

A_s / R = max over the columns that stay on the int8 path.  Route rule: smallest P in {5, 6} with SQ_P <= TAU_SQ and X_P <= tau_x(tokens)
after sending at most JMAX columns to the fp64 column kernel (greedy: the column whose removal lowers the violation most);
otherwise the whole statistic goes to the fp64 kernel.  ||N_j|| enters through the integer lower bound
2^32 (||256 d_0 + d_1|| - sqrt(nnz_j) / 2), so every decision is a function of integers and bit-reproducible.
"""
import math

import numpy as np

NP_, TOP_SHIFT = 6, 38
TAU_SQ, TAU_X, JMAX = 1e-12, 1e-11, 32            # TAU_X: the cross-term threshold of a call of >= 10240 tokens


def tau_x_of(tokens):
    """The threshold on X_P grows with the EFFECTIVE token count -- the fewest nonzero elements any column has, at most the tokens
    of the call (cov_i8.hip tau_x_of): for uncorrelated columns the measured error sits ~4.5 / sqrt(tokens) below X_P, an
    averaging a short call or a sparse column does not have.  1e-12 up to 1024 tokens, 1e-11 from 10240."""
    return min(TAU_X, max(1e-12, 1e-12 * tokens / 1024.0))


PAIRS = {P: ([(s, s) for s in range(NP_) if 2 * s >= P], [(s, t) for s in range(NP_) for t in range(NP_) if s != t and s + t >= P])
         for P in (5, 6)}


def bf16_parts(X):
    """torch bf16 [T, n] -> (sig int64 in [-255, 255], effective exponent >= 1); value = sig 2^(ee - 134)."""
    import torch
    bits = X.view(torch.int16).to(torch.int32).numpy().astype(np.int64) & 0xFFFF
    e, m = (bits >> 7) & 0xFF, bits & 0x7F
    sig = np.where(e > 0, m + 128, m) * np.where(bits >> 15, -1, 1)
    return sig, np.maximum(e, 1)


def digits(X):
    """-> (d [6, T, n] int64 balanced digits, E [n], N [T, n], rounded [n], nnz [n])"""
    sig, ee = bf16_parts(X)
    E = np.where(sig != 0, ee, 1).max(axis=0)
    sh = E[None, :] - ee
    up = sig << np.clip(TOP_SHIFT - sh, 0, None)
    dn = np.clip(sh - TOP_SHIFT, 1, 62)
    mag = np.where(dn > 9, 0, (np.abs(sig) + (1 << (dn - 1))) >> dn)
    N = np.where(sh <= TOP_SHIFT, up, np.sign(sig) * mag)
    rounded = ((sh > TOP_SHIFT) & (sig != 0)).sum(0)
    ds, R = [], N.copy()
    for _ in range(NP_ - 1):
        b = ((R + 128) & 0xFF) - 128
        ds.append(b)
        R = (R - b) >> 8
    ds.append(R)
    return np.stack(ds[::-1]), E, N, rounded, (sig != 0).sum(0)


def column_stats(d, rounded, nnz):
    """The integers the split pass accumulates per column: q[s] = sum d_s^2 (s = 0..5), q[6] = sum d_0 d_1."""
    q = np.concatenate([(d * d).sum(1), (d[0] * d[1]).sum(0)[None]]).astype(np.int64)
    return {"q": q, "rounded": np.asarray(rounded, np.int64), "nnz": np.asarray(nnz, np.int64)}


def alphas(st):
    """-> (alpha [6, n], rho [n]) in fp64 from the integer statistics, with the integer lower bound on ||N_j||."""
    q = st["q"].astype(np.float64)
    hi2 = 65536.0 * q[0] + 512.0 * q[6] + q[1]
    norm = (np.sqrt(np.maximum(hi2, 0.0)) - 0.5 * np.sqrt(st["nnz"].astype(np.float64))) * 4294967296.0
    # (norm <= 0 can only happen for a column of denormals, which has nothing below plane 1; 1e300 keeps the test conservative)
    inv = np.where(st["nnz"] > 0, np.where(norm > 0, 1.0 / np.where(norm > 0, norm, 1.0), 1e300), 0.0)
    a = np.stack([np.where(q[s] > 0, np.sqrt(q[s]) * (256.0 ** (NP_ - 1 - s)) * inv, 0.0) for s in range(NP_)])
    return a, np.where(st["rounded"] > 0, 0.5 * np.sqrt(st["rounded"].astype(np.float64)) * inv, 0.0)


def terms(A, R, P):
    sq = sum(A[s] * A[t] for s, t in PAIRS[P][0]) + 2.0 * R + R * R
    x = sum(A[s] * A[t] for s, t in PAIRS[P][1])
    return sq, x


def violation(A, R, P, tau_x=TAU_X):
    sq, x = terms(A, R, P)
    return max(sq / TAU_SQ, x / tau_x)


def entry_bound(a, rho, P):
    """[n, n] bound on |err_ij| / sqrt(sigma_ii sigma_jj)."""
    b = sum(a[s][:, None] * a[t][None, :] for s in range(NP_) for t in range(NP_) if s + t >= P)
    return b + rho[:, None] + rho[None, :] + rho[:, None] * rho[None, :]


def route(st, jmax=JMAX, sort=True, tokens=1 << 20, tolerance=1.0):
    """-> (planes: 5, 6 or 0 = fp64 kernel for the whole statistic; sorted list of columns for the fp64 column kernel;
    (SQ, X) of the columns that stay).  tolerance: mdg_cov_i8_set_tolerance's factor on both thresholds."""
    a, rho = alphas(st)
    tau_x = tau_x_of(tokens)
    vals = np.concatenate([a, rho[None]])            # 7 quantities per column
    n = vals.shape[1]
    for P in (5, 6):
        # prefilter: whatever jmax columns leave, the (jmax + 1)-th largest of every quantity stays
        if n > jmax:
            floor = np.partition(vals, n - 1 - jmax, axis=1)[:, n - 1 - jmax]
            if violation(floor[:6], floor[6], P, tau_x) > tolerance:
                continue
        live = np.ones(n, bool)
        out = []
        while True:
            masked = np.where(live[None, :], vals, -1.0)
            arg1 = masked.argmax(axis=1)             # (lowest index among equals)
            max1 = masked[np.arange(7), arg1]
            m2 = masked.copy()
            m2[np.arange(7), arg1] = -1.0
            max2 = np.maximum(m2.max(axis=1), 0.0)
            max1 = np.maximum(max1, 0.0)
            if violation(max1[:6], max1[6], P, tau_x) <= tolerance:
                return P, (sorted(out) if sort else out), terms(max1[:6], max1[6], P)
            if len(out) == jmax or not live.any():
                break
            best, best_v = -1, math.inf
            for q in range(7):                       # candidates: the columns that hold a maximum, in quantity order
                c = int(arg1[q])
                A2 = np.where(arg1 == c, max2, max1)
                v = violation(A2[:6], A2[6], P, tau_x)
                if v < best_v:
                    best, best_v = c, v
            live[best] = False
            out.append(best)
    return 0, [], (math.inf, math.inf)


def product(d, E, P):
    """The modelled P-plane product as fp64 (sum over kept plane pairs, integer class sums folded high to low)."""
    n = d.shape[2]
    df = d.astype(np.float64)                # |class sum| < 2^53: the fp64 products below are exact integers
    cls = [np.zeros((n, n)) for _ in range(2 * NP_ - 1)]
    for s in range(NP_):
        for t in range(NP_):
            if s + t < P:
                cls[s + t] += df[s].T @ df[t]
    acc = np.zeros((n, n))
    for k in range(P - 1, -1, -1):
        acc += np.ldexp(cls[k], 80 - 8 * k)
    sc = np.ldexp(1.0, (E - 172).astype(np.int64))
    return acc * sc[:, None] * sc[None, :]


LO_CHUNK_TOKENS, LO_CAP, EXACT_ROUNDING = 64 * 32, 128, 5e-15      # cov_i8.hip: LO_CHUNK_STEPS x KS, LO_CAP; modegpt_hip.h: MDG_I8_EXACT_ROUNDING


def list_counts(lo):
    """lo [T, n] bool (element has L != 0) -> [segments, n]: the lengths of the exact route's event lists -- one per (2048-token
    segment, column) (i8_extract_lo_kernel)."""
    T, n = lo.shape
    pad = (-T) % LO_CHUNK_TOKENS
    if pad:
        lo = np.concatenate([lo, np.zeros((pad, n), bool)])
    return lo.reshape(-1, LO_CHUNK_TOKENS, n).sum(axis=1)


def remainder_counts(d):
    """Lengths of the event lists of the digits d [6, T, n]: elements with a nonzero digit in planes 3 .. 5."""
    return list_counts((d[3] != 0) | (d[4] != 0) | (d[5] != 0))


def route_of(X, chunk=1024, tolerance=1.0, offer_exact=True):
    """Route of a whole bf16 activation matrix (torch CPU tensor [T, n]): the statistics are taken in column chunks (int64
    temporaries of a 32768 x 14336 batch would not fit), the decision over all columns.  -> dict like ops.cov_accum_i8's
    route_info: planes, columns (in the order the greedy took them), sq, x, bound, exact.  exact (the exact route: nine plane
    pairs + the fp64 remainder products, cov_i8.hip "the exact route"): the route kernel's decision stands, and when no event list
    of the columns that stayed overflows its segment the truncated product is replaced -- the bound is then the rounded-element
    term 2 R + R^2 plus fp64 rounding.  offer_exact: True (MDG_I8_EXACT_ALWAYS), False (MDG_I8_NO_EXACT) or "auto" (flags 0: only
    for the six-plane class)."""
    qs, rs, ns, lo = [], [], [], []
    for c0 in range(0, X.shape[1], chunk):
        d, _, _, rounded, nnz = digits(X[:, c0:c0 + chunk].contiguous())
        st = column_stats(d, rounded, nnz)
        qs.append(st["q"]); rs.append(st["rounded"]); ns.append(st["nnz"])
        lo.append((d[3] != 0) | (d[4] != 0) | (d[5] != 0))
    st = {"q": np.concatenate(qs, axis=1), "rounded": np.concatenate(rs), "nnz": np.concatenate(ns)}
    nz = st["nnz"][st["nnz"] > 0]
    planes, cols, (sq, x) = route(st, sort=False, tokens=min(X.shape[0], int(nz.min()) if nz.size else X.shape[0]), tolerance=tolerance)
    exact = False
    if offer_exact == "auto":          # the library's default: the exact route where it is the faster product -- the six-plane class,
        offer_exact = planes == 6 or X.shape[1] >= 4096     # and large five-plane statistics (cov_i8.hip lo_offered)
    if planes and offer_exact and X.shape[1] % 32 == 0:
        lo = np.concatenate(lo, axis=1)
        lo[:, cols] = False                            # the digits of the columns that left are cleared before the lists are made
        n = lo.shape[1]
        exact = bool(list_counts(lo).max() <= LO_CAP)
        if exact:
            _, rho = alphas(st)
            live = np.ones(n, bool)
            live[cols] = False
            R = float(rho[live].max()) if live.any() else 0.0
            sq, x = 2.0 * R + R * R + EXACT_ROUNDING, 0.0
    return {"planes": planes, "columns": cols, "sq": sq, "x": x, "bound": sq + x, "exact": exact, "stats": st}
