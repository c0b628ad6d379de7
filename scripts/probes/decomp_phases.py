"""Where the decomposition of one Llama-3-8B layer spends its time: each phase of the MLP chain timed alone with HIP events, plus
the fp64 GEMM kernel's rate at the shapes the blocked Cholesky / substitution issue (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops, _lib

dev = torch.device("cuda:0")
F64 = torch.float64


def timeit(fn, n=2, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


n, d, r = 14336, 4096, 10035
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(2 * n, n, device=dev, generator=g, dtype=F64)
S = torch.empty(n, n, device=dev, dtype=F64)
ops.gemm(X, X, S, alpha=1.0 / (2 * n), trans_a=True)
del X
lib = _lib.load()


def padded(m):
    """[m, m] view of a buffer whose row pitch is m rounded up to 16 doubles (what mdg_nystrom_down gives C_kk: an odd pitch puts the
    GEMMs on their element-wise staging path)."""
    return torch.empty(m, (m + 15) // 16 * 16, device=dev, dtype=F64)[:, :m]


def potrf(m):
    A = padded(m); A.copy_(S[:m, :m])
    A.diagonal().add_(1e-4)
    t = timeit(lambda: (A.copy_(S[:m, :m]), A.diagonal().add_(1e-4), ops.potrf_lower(A)), n=2)
    tc = timeit(lambda: (A.copy_(S[:m, :m]), A.diagonal().add_(1e-4)), n=2)
    t -= tc
    print(f"potrf_lower n={m}: {t*1e3:.1f} ms  {m**3/3/t/1e12:.1f} TF")
    return A


which = sys.argv[1:] or ["phases", "gemm"]
if "potrfonly" in which:
    potrf(n)
if "inverseonly" in which:      # (for a kernel trace of the triangular inverse alone)
    A = S.clone(); A.diagonal().add_(1e-4)
    inv = ops.potrf_lower(A)
    out = torch.empty(n, device=dev, dtype=F64)
    nbytes = lib.mdg_chol_inverse_diag_ws_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: lib.mdg_chol_inverse_diag(A.data_ptr(), n, n, inv.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, st), n=2)
    print(f"chol_inverse_diag n={n}: {t*1e3:.1f} ms  {n**3/3/t/1e12:.1f} TF (n^3/3)")
if "potrsonly" in which:        # (for a kernel trace of the Nystrom substitution alone)
    Ar = padded(r); Ar.copy_(S[:r, :r]); Ar.diagonal().add_(1e-4)
    invr = ops.potrf_lower(Ar)
    B = torch.randn(r, d, device=dev, generator=g, dtype=F64)
    t = timeit(lambda: ops.potrs_lower(Ar, invr, B), n=2)
    print(f"potrs_lower n={r} nrhs={d}: {t*1e3:.1f} ms  {2*r*r*d/t/1e12:.1f} TF")
if "phases" in which:
    potrf(n)
    A = S.clone(); A.diagonal().add_(1e-4)
    inv = ops.potrf_lower(A)
    out = torch.empty(n, device=dev, dtype=F64)
    nbytes = lib.mdg_chol_inverse_diag_ws_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: lib.mdg_chol_inverse_diag(A.data_ptr(), n, n, inv.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, st), n=2)
    print(f"chol_inverse_diag n={n}: {t*1e3:.1f} ms  {n**3/3/t/1e12:.1f} TF (n^3/3)")
    del ws
    potrf(r)
    Ar = padded(r); Ar.copy_(S[:r, :r]); Ar.diagonal().add_(1e-4)
    invr = ops.potrf_lower(Ar)
    B = torch.randn(r, d, device=dev, generator=g, dtype=F64)
    t = timeit(lambda: ops.potrs_lower(Ar, invr, B), n=2)
    print(f"potrs_lower n={r} nrhs={d}: {t*1e3:.1f} ms  {2*r*r*d/t/1e12:.1f} TF")
    Wd = (torch.randn(d, n, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    idx = torch.sort(torch.randperm(n, device=dev)[:r]).values
    Xc = torch.empty(r, d, device=dev, dtype=F64)
    t = timeit(lambda: ops.gemm(S, Wd, Xc, trans_b=True, a_rows=idx), n=2)
    print(f"cross term [{r} x {n}] gathered rows x W_d^T [{n} x {d}]: {t*1e3:.1f} ms  {2*r*n*d/t/1e12:.1f} TF")
    t = timeit(lambda: ops.gemm(S, Wd.to(F64), Xc, trans_b=True, a_rows=idx), n=2)
    print(f"   the same with W_d widened to fp64 first (the conversion included): {t*1e3:.1f} ms")
    Wt = Wd.to(F64).t().contiguous()
    t = timeit(lambda: ops.gemm(S, Wt, Xc, a_rows=idx), n=2)
    print(f"   W_d^T as a row-major fp64 [n x d] matrix (conversion and transpose NOT included): {t*1e3:.1f} ms")
if "gemm" in which:
    A = torch.randn(n, n, device=dev, generator=g, dtype=F64)
    C = torch.zeros(n, n, device=dev, dtype=F64)
    for (M, N, K, flags, ta, tb, what) in (
            (13312, 13312, 1024, _lib.MDG_GEMM_LOWER_ONLY, False, True, "outer rank-1024 update, lower only (A A^T)"),
            (7168, 7168, 1024, _lib.MDG_GEMM_LOWER_ONLY, False, True, "outer rank-1024 update, lower only, half way"),
            (13312, 896, 128, 0, False, True, "inner rank-128 update of an outer panel"),
            (7168, 448, 128, 0, False, True, "inner rank-128 update, half way"),
            (13312, 128, 128, 0, False, False, "panel solve (x inverse block)"),
            (8192, 4096, 1024, 0, False, False, "substitution: rank-1024 carry, nrhs 4096"),
            (896, 4096, 128, 0, False, False, "substitution: inner rank-128, nrhs 4096"),
            (128, 4096, 128, 0, False, False, "substitution: diagonal block x rhs"),
            (8192, 8192, 8192, 0, False, False, "square 8192"),
    ):
        a = A[:K, :M].T if ta else A[:M, :K]
        b = A[:N, :K].T if tb else A[:K, :N]
        c = C[:M, :N]
        t = timeit(lambda: ops.gemm(a, b, c, alpha=-1.0, beta=1.0, flags=flags), n=5, warm=2)
        fl = 2.0 * M * N * K * (0.5 if flags & _lib.MDG_GEMM_LOWER_ONLY else 1.0)
        print(f"gemm {M}x{N}x{K} {what}: {t*1e6:.0f} us  {fl/t/1e12:.1f} TF")
if "async" in which:
    # one layer's whole compress_nystrom + compress_qk + compress_vo chain in deferred-status mode.  The host
    # must come back long before the device is done (no synchronisation inside the layer), and a not-positive-definite
    # input must still surface as LinAlgError -- when the status is read.
    import time
    from modegpt_amd import engine
    shape = engine.SHAPES["llama-3-8b"]
    w = engine.make_layer_weights(shape, 1234, dev)
    batch = engine.make_activation_batch(shape, 8192, seed=5, device=dev)
    covs = engine.new_covs(shape, dev)
    engine.accumulate(covs, batch, shape)
    engine.finalize(covs, 4)
    ad = engine.TensorAdapter(shape, {0: w})
    engine.compress_layer(ad, 0, covs, 0.7)                     # warm-up (checked)
    for mode in ("deferred", "checked"):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        engine.compress_layer(ad, 0, covs, 0.7, check=(mode == "checked"))
        e1.record(); t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        ad.check_chains()
        print(f"layer chain, {mode}: host back after {t_host*1e3:.1f} ms, device busy {e0.elapsed_time(e1):.1f} ms "
              f"({'no host synchronisation inside the layer' if t_host * 1e3 < 0.5 * e0.elapsed_time(e1) else 'the host waited'})")
    bad = {k: v.clone() for k, v in covs.items()}
    bad["mlp"][7, 7] = -1.0
    try:
        engine.compress_layer(ad, 0, bad, 0.7, check=False)
        print("not-PD sigma_mlp: the chain enqueued without raising (deferred) ...")
        ad.check_chains()
        print("... and check_chains() did NOT raise: BUG")
    except torch.linalg.LinAlgError as e:
        print("... check_chains() raised LinAlgError:", str(e)[:110])
if "pair" in which:
    # two (three, four) layers' chains: back to back on one stream against side by side on a stream each.  A chain is a string of short
    # dependent kernels between large GEMMs; another layer's chain is independent of it.
    import time
    from modegpt_amd import engine
    shape = engine.SHAPES["llama-3-8b"]
    nl = 4
    ws = {i: engine.make_layer_weights(shape, 1234 + i, dev) for i in range(nl)}
    batch = engine.make_activation_batch(shape, 8192, seed=5, device=dev)
    covs = engine.new_covs(shape, dev)
    engine.accumulate(covs, batch, shape)
    engine.finalize(covs, 4)
    ad = engine.TensorAdapter(shape, ws)
    engine.compress_layer(ad, 0, covs, 0.7)                     # warm-up (checked)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nl)]
    main = torch.cuda.current_stream(dev)
    for k in (1, 2, 3, 4):
        for rep in range(2):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for i in range(k):
                streams[i].wait_stream(main)
                with torch.cuda.stream(streams[i]):
                    engine.compress_layer(ad, i, covs, 0.7, check=False)
            for i in range(k):
                main.wait_stream(streams[i])
            e1.record(main)
            torch.cuda.synchronize()
            ad.check_chains()
        print(f"{k} chain(s) side by side: {e0.elapsed_time(e1):.1f} ms = {e0.elapsed_time(e1) / k:.1f} ms per layer")
