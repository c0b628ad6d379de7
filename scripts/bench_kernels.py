"""Per-kernel timing at Llama-3-8B shapes (dev tool; the judged numbers come from bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modegpt_amd import ops, _lib

dev = torch.device("cuda:0")
F64 = torch.float64

def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

print("device", ops.device_info(0))
print("fp64 MFMA probe TFLOP/s:", ops.probe_mfma_f64(8192))
which = sys.argv[1:] or ["cov", "mlp", "vo", "qk"]
T, d_ff, d, nh, nkv, hd = 32768, 14336, 4096, 32, 8, 128
g = torch.Generator(device=dev).manual_seed(0)
def acts(t, f):
    c = torch.exp(torch.empty(f, device=dev).uniform_(-3.0, 0.7, generator=g))
    return (torch.randn(t, f, device=dev, generator=g) * c).to(torch.bfloat16)
if "cov" in which:
    H = acts(T, d_ff); S = torch.zeros(d_ff, d_ff, dtype=F64, device=dev)
    t = timeit(lambda: ops.cov_accum(S, H), n=2)
    fl = T * d_ff * (d_ff + 1)
    print(f"cov mlp  {T}x{d_ff}: {t*1e3:.1f} ms  {fl/t/1e12:.1f} TF (syrk count)")
    X = acts(T, d); Sx = torch.zeros(d, d, dtype=F64, device=dev)
    t = timeit(lambda: ops.cov_accum(Sx, X), n=3)
    print(f"cov x    {T}x{d}: {t*1e3:.1f} ms  {T*d*(d+1)/t/1e12:.1f} TF")
    Q = acts(T, nh * hd); Sq = torch.zeros(nh, hd, hd, dtype=F64, device=dev)
    t = timeit(lambda: ops.cov_accum(Sq, Q, n_heads=nh), n=3)
    print(f"cov q    {T}x{nh}x{hd}: {t*1e3:.1f} ms  {T*nh*hd*(hd+1)/t/1e12:.1f} TF")
    K = acts(T, nkv * hd); Sk = torch.zeros(nkv, hd, hd, dtype=F64, device=dev)
    t = timeit(lambda: ops.cov_accum(Sk, K, n_heads=nkv), n=3)
    print(f"cov k    {T}x{nkv}x{hd}: {t*1e3:.1f} ms  {T*nkv*hd*(hd+1)/t/1e12:.1f} TF")
    t = timeit(lambda: ops.cov_accum_multi([(S, H, 1), (Sx, X, 1), (Sq, Q, nh), (Sk, K, nkv)]), n=3)
    fl4 = T * (d_ff * (d_ff + 1) + d * (d + 1) + (nh + nkv) * hd * (hd + 1))
    print(f"cov fused (mlp+x+q+k, one launch): {t*1e3:.1f} ms  {fl4/t/1e12:.1f} TF")
    t = timeit(lambda: ops.cov_finalize(S, 1.0 / T), n=2)
    print(f"finalize {d_ff}: {t*1e3:.2f} ms")
    if "mlp" in which:
        ops.cov_finalize(Sx, 1.0 / T); ops.cov_finalize(Sq, 1.0 / T); ops.cov_finalize(Sk, 1.0 / T)
        lam = float(torch.tensor(1e-4, dtype=torch.float32).double())
        t0 = time.time(); sc = ops.ridge_scores(S, lam); torch.cuda.synchronize(); print(f"ridge_scores first {time.time()-t0:.3f}s")
        t = timeit(lambda: ops.ridge_scores(S, lam), n=2, warm=0); print(f"ridge_scores {d_ff}: {t*1e3:.1f} ms")
        r = int(d_ff * 0.7)
        t = timeit(lambda: ops.select_smallest_sorted(sc, r), n=3); print(f"select: {t*1e3:.2f} ms")
        idx = ops.select_smallest_sorted(sc, r)
        Wd = (torch.randn(d, d_ff, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        Wu = (torch.randn(d_ff, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        t = timeit(lambda: ops.nystrom_down(S, idx, Wd), n=2); print(f"nystrom_down r={r}: {t*1e3:.1f} ms")
        t = timeit(lambda: ops.gather_rows(Wu, idx), n=3); print(f"gather up: {t*1e3:.2f} ms")
    if "vo" in which:
        Wv = (torch.randn(nkv * hd, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        Wo = (torch.randn(d, nh * hd, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        t = timeit(lambda: ops.vo_compress(Sx, Wv, Wo, nh, nkv, hd, 88, 1e-5), n=2); print(f"vo_compress gqa: {t*1e3:.1f} ms")
        Wv2 = (torch.randn(nh * hd, d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        t = timeit(lambda: ops.vo_compress(Sx, Wv2, Wo, nh, nh, hd, 88, 1e-5), n=2); print(f"vo_compress mha: {t*1e3:.1f} ms")
    if "qk" in which:
        t = timeit(lambda: ops.qk_select(Sq, Sk, 88, _lib.MDG_QK_ROPE_GROUPED, 1e-4, 1e-2), n=3); print(f"qk_select: {t*1e3:.3f} ms")
if "cov1r" in which:
    # one exactly-resident round: T=31 -> 496 tiles on 512 slots; isolates in-loop efficiency from tail effects
    n = 128 * 31
    X = acts(131072, n); S1 = torch.zeros(n, n, dtype=F64, device=dev)
    t = timeit(lambda: ops.cov_accum(S1, X), n=3)
    ex = 496 * 128 * 128 * 2 * 131072
    print(f"cov 1-round {131072}x{n}: {t*1e3:.1f} ms  {131072*n*(n+1)/t/1e12:.1f} TF syrk, executed {ex/t/1e12:.1f} TF on 496/512 slots -> {ex/t/1e12*512/496:.1f} TF-equivalent")
if "sqrt" in which:
    import time
    X = acts(16384, 4096).double(); M = X.T @ X / 16384
    t0 = time.time(); r, ri, lam = ops.sqrt_psd_large(M, 1e-5, False, True); torch.cuda.synchronize()
    print(f"sqrt_psd_large n=4096: {time.time()-t0:.2f} s; check ||r r - (M + rho I)||/||M|| = {((r @ r - M - 1e-5*torch.eye(4096, device=dev, dtype=F64)).norm()/M.norm()).item():.2e}")
if "rope" in which:
    # compressed Llama-3-8B attention, one calibration-sized batch: 16 x 2048 tokens, 32 q heads / 8 kv heads, 88 of 128 kept
    B, Tt = 16, 2048
    for name, heads, norm, r in (("q", nh, False, 88), ("k", nkv, False, 88), ("q+norm", nh, True, 88),
                                 ("q r=76", nh, False, 76), ("q r=102", nh, False, 102), ("q r=128", nh, False, 128)):
        x = torch.randn(B, Tt, heads * r, device=dev, generator=g).to(torch.bfloat16)
        ang = torch.rand(1, Tt, hd // 2, device=dev, generator=g) * 6.28
        emb = torch.cat((ang, ang), -1)
        cos, sin = emb.cos().to(torch.bfloat16), emb.sin().to(torch.bfloat16)
        idx = torch.stack([torch.randperm(hd // 2, device=dev)[:r // 2] for _ in range(nkv)])
        mask = torch.cat((idx, idx + hd // 2), dim=1)
        w = torch.ones(hd, device=dev, dtype=torch.bfloat16) if norm else None
        t = timeit(lambda: ops.rope_gather(x, cos, sin, mask, heads, nkv if heads == nh else heads, hd, norm_weight=w), n=20, warm=3)
        nbytes = 2 * x.numel() * 2
        print(f"rope_gather {name:7s} [{B},{Tt},{heads}x{r}] bf16: {t*1e6:.1f} us  {nbytes/t/1e9:.0f} GB/s (algorithmic: read + write once)")
        if not norm:   # the eager chain of the reference's modeling file on the same GPU, for scale
            mq = mask.repeat_interleave(heads // mask.shape[0], 0) if heads == nh else mask[:heads]
            def eager():
                q = x.view(B, Tt, heads, r).transpose(1, 2)
                c = cos[:, :, mq].permute(0, 2, 1, 3); s = sin[:, :, mq].permute(0, 2, 1, 3)
                h2 = r // 2
                return q * c + torch.cat((-q[..., h2:], q[..., :h2]), -1) * s
            te = timeit(eager, n=10, warm=2)
            print(f"   torch eager chain: {te*1e6:.1f} us  ({te/t:.1f}x)")
if "sqrt" in which:
    n = 4096
    X = acts(4 * n, n).double()
    M = X.T @ X / (4 * n)
    for ev in (False, True):
        t0 = time.time(); r = ops.sqrt_psd_large(M, 1e-5, False, True, want_evals=ev); torch.cuda.synchronize()
        t = time.time() - t0
        err = ((r[0] @ r[0] - M - 1e-5 * torch.eye(n, dtype=F64, device=dev)).abs().max() / M.abs().max()).item()
        ierr = ((r[0] @ r[1] - torch.eye(n, dtype=F64, device=dev)).abs().max()).item()
        print(f"sqrt_psd_large n={n} {'block Jacobi (eigenvalues)' if ev else 'Newton-Schulz (GEMM only)'}: {t:.3f} s  "
              f"|R R - A|/|A| = {err:.1e}  |R R^-1 - I| = {ierr:.1e}")
if "cov64" in which:
    # is the in-loop bf16 -> fp64 conversion worth hoisting into a pre-pass?  same problem, operands already fp64 in HBM
    H = acts(T, d_ff); S = torch.zeros(d_ff, d_ff, dtype=F64, device=dev)
    t = timeit(lambda: ops.cov_accum(S, H), n=3)
    fl = T * d_ff * (d_ff + 1)
    print(f"cov mlp bf16 input {T}x{d_ff}: {t*1e3:.1f} ms  {fl/t/1e12:.1f} TF")
    t0 = timeit(lambda: H.double(), n=3)
    H64 = H.double()
    t = timeit(lambda: ops.cov_accum(S, H64), n=3)
    print(f"cov mlp fp64 input {T}x{d_ff}: {t*1e3:.1f} ms  {fl/t/1e12:.1f} TF   (+ torch bf16->fp64 pre-pass {t0*1e3:.2f} ms)")
if "covi8" in which:
    H = acts(T, d_ff)
    S8 = torch.zeros(d_ff, d_ff, dtype=F64, device=dev); S64 = torch.zeros_like(S8)
    used = ops.cov_accum_i8(S8, H); ops.cov_accum(S64, H)
    low = torch.tril(torch.ones(d_ff, d_ff, dtype=torch.bool, device=dev))
    print(f"cov_accum_i8 used the int8 route: {used};  max |i8 - fp64| / max |sigma| = {((S8 - S64)[low].abs().max() / S64.abs().max()).item():.2e}")
    t8 = timeit(lambda: ops.cov_accum_i8(S8, H), n=3)
    t64 = timeit(lambda: ops.cov_accum(S64, H), n=3)
    fl = T * d_ff * (d_ff + 1)
    print(f"cov mlp {T}x{d_ff}: int8 digit planes {t8*1e3:.1f} ms ({fl/t8/1e12:.1f} fp64-SYRK-equivalent TF)   fp64 MFMA {t64*1e3:.1f} ms ({fl/t64/1e12:.1f} TF)")
    X = acts(T, d); Sx = torch.zeros(d, d, dtype=F64, device=dev)
    t8 = timeit(lambda: ops.cov_accum_i8(Sx, X), n=3); t64 = timeit(lambda: ops.cov_accum(Sx, X), n=3)
    print(f"cov x   {T}x{d}: int8 digit planes {t8*1e3:.2f} ms   fp64 MFMA {t64*1e3:.2f} ms")
if "covi8fused" in which:
    # the four statistics of a Llama-3-8B calibration batch in ONE int8 launch (what the hooks enqueue): sigma_mlp, sigma_x, and
    # the per-head sigma_q / sigma_k as diagonal tiles of the same tile schedule
    H, X, Q, K = acts(T, d_ff), acts(T, d), acts(T, nh * hd), acts(T, nkv * hd)
    S = [torch.zeros(d_ff, d_ff, dtype=F64, device=dev), torch.zeros(d, d, dtype=F64, device=dev),
         torch.zeros(nh, hd, hd, dtype=F64, device=dev), torch.zeros(nkv, hd, hd, dtype=F64, device=dev)]
    items = [(S[0], H, 1), (S[1], X, 1), (S[2], Q, nh), (S[3], K, nkv)]
    used = ops.cov_accum_i8_multi(items, report=True)
    t8 = timeit(lambda: ops.cov_accum_i8_multi(items), n=4)
    tsep = timeit(lambda: (ops.cov_accum_i8(S[0], H, report=False), ops.cov_accum_i8(S[1], X, report=False),
                           ops.cov_accum(S[2], Q, n_heads=nh), ops.cov_accum(S[3], K, n_heads=nkv)), n=4)
    print(f"cov fused int8 launch (mlp + x + q + k, route {used}): {t8*1e3:.2f} ms   the same four as separate launches (two int8, two fp64): {tsep*1e3:.2f} ms")
if "covi8p6" in which:
    # SiLU-gated activations (the MLP statistic of a real Llama): the depth statistic picks six planes
    gg = torch.randn(T, d_ff, device=dev, generator=g); uu = torch.randn(T, d_ff, device=dev, generator=g)
    H = (torch.nn.functional.silu(gg) * uu).to(torch.bfloat16); del gg, uu
    S8 = torch.zeros(d_ff, d_ff, dtype=F64, device=dev); S64 = torch.zeros_like(S8)
    used = ops.cov_accum_i8(S8, H); ops.cov_accum(S64, H)
    dd = torch.sqrt(torch.diag(S64))
    low = torch.tril(torch.ones(d_ff, d_ff, dtype=torch.bool, device=dev))
    err = 0.0
    for i0 in range(0, d_ff, 2048):   # entry-wise error over sqrt(sigma_ii sigma_jj), in row blocks to bound memory
        blk = ((S8[i0:i0 + 2048] - S64[i0:i0 + 2048]).abs() / (dd[i0:i0 + 2048, None] * dd[None])) * low[i0:i0 + 2048]
        err = max(err, blk.max().item())
    print(f"SiLU-gated columns: route {used} planes;  max |i8 - fp64| / sqrt(s_ii s_jj) = {err:.2e}")
    t8 = timeit(lambda: ops.cov_accum_i8(S8, H), n=3)
    t64 = timeit(lambda: ops.cov_accum(S64, H), n=3)
    print(f"cov mlp {T}x{d_ff} (SiLU-gated): int8 digit planes {t8*1e3:.1f} ms   fp64 MFMA {t64*1e3:.1f} ms")
if "covi8massive" in which:
    # four BOS-like columns (bulk 12-15 binades under three spikes) in sigma_mlp- and sigma_x-sized statistics: the route hands
    # them to the fp64 column kernel; whole-call time against the clean batch, and the route the device reports
    def massive(X, cols):
        X = X.clone()
        for i, c in enumerate(cols):
            top = X[:, c].float().abs().max()
            X[:, c] = (X[:, c].float() * 2.0 ** -(12 + i)).to(torch.bfloat16)
            X[torch.randperm(X.shape[0], device=dev, generator=g)[:3], c] = (top * 1.5).to(torch.bfloat16)
        return X
    for n_, name in ((d_ff, "mlp"), (d, "x")):
        H = acts(T, n_); Hm = massive(H, [5, 129, n_ // 2 + 77, n_ - 1])
        S8 = torch.zeros(n_, n_, dtype=F64, device=dev)
        info, info_m = {}, {}
        ops.cov_accum_i8(S8, H, route_info=info); ops.cov_accum_i8(S8, Hm, route_info=info_m)
        t0 = timeit(lambda: ops.cov_accum_i8(S8, H, report=False), n=5, warm=2)
        t1 = timeit(lambda: ops.cov_accum_i8(S8, Hm, report=False), n=5, warm=2)
        t0b = timeit(lambda: ops.cov_accum_i8(S8, H, report=False), n=5, warm=2)
        print(f"cov {name} {T}x{n_}: clean {t0*1e3:.3f} / {t0b*1e3:.3f} ms (planes {info['planes']}, bound {info['bound']:.2e});  4 massive columns "
              f"{t1*1e3:.3f} ms (+{100 * (t1 / min(t0, t0b) - 1):.1f} %), planes {info_m['planes']}, columns {info_m['columns']}, bound {info_m['bound']:.2e}")
