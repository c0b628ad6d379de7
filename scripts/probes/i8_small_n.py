import torch, time, sys
sys.path.insert(0, '.')
from modegpt_amd import ops
dev = torch.device('cuda:0')
for n in (1024, 1536, 2048, 2560, 3072, 4096, 5120):
    x = torch.randn(32768, n, device=dev).to(torch.bfloat16)
    s = torch.zeros(n, n, dtype=torch.float64, device=dev)
    res = []
    for fn in (lambda: ops.cov_accum_i8(s, x), lambda: ops.cov_accum(s, x)):
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        res.append((time.time() - t0) * 100)
    print(f"n={n}: i8 {res[0]:.2f} ms  f64 {res[1]:.2f} ms")
