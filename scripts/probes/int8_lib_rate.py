import torch, time
dev = torch.device("cuda:0")
n, T = 14336, 32768
A = torch.randint(-128, 127, (n, T), dtype=torch.int8, device=dev)
Bm = A.t().contiguous()  # [T, n]
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for name, fn in (("int_mm A @ A^T (B given as [T,n] contiguous)", lambda: torch._int_mm(A, Bm)),
                 ("int_mm A @ A.t() view", lambda: torch._int_mm(A, A.t()))):
    try:
        t = timeit(fn)
        print(f"{name}: {t*1e3:.1f} ms  {2*n*n*T/t/1e12:.0f} TOPS (full GEMM count)")
    except Exception as e:
        print(name, "failed:", type(e).__name__, str(e)[:200])
Ab = torch.randn(n, T, device=dev).to(torch.bfloat16)
t = timeit(lambda: Ab @ Ab.t())
print(f"bf16 A @ A^T: {t*1e3:.1f} ms  {2*n*n*T/t/1e12:.0f} TFLOPS")
