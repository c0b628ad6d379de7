"""One int8 launch for all four statistics of a Llama-3-8B layer and batch, against sigma_mlp in a launch of its own + the three small
ones together (what ops.cov_accum_multi does).   python scripts/probes/fused_all_four.py [gaussian|silu_gated]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import ops
dev = torch.device("cuda:0"); F64 = torch.float64
kind = sys.argv[1] if len(sys.argv) > 1 else "gaussian"
T = 32768
def gaussian(n, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    c = torch.exp(torch.empty(n, device=dev).uniform_(math.log(0.05), math.log(2.0), generator=g))
    return (torch.randn(T, n, device=dev, generator=g) * c).to(torch.bfloat16)
def gated(n, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.nn.functional.silu(torch.randn(T, n, device=dev, generator=g)); a.mul_(torch.randn(T, n, device=dev, generator=g))
    return a.to(torch.bfloat16)
mlp = (torch.zeros(14336, 14336, dtype=F64, device=dev), (gated if kind == "silu_gated" else gaussian)(14336, 1), 1)
x = (torch.zeros(4096, 4096, dtype=F64, device=dev), gaussian(4096, 2), 1)
q = (torch.zeros(32, 128, 128, dtype=F64, device=dev), gaussian(4096, 3), 32)
k = (torch.zeros(8, 128, 128, dtype=F64, device=dev), gaussian(1024, 4), 8)
def split():
    ops.cov_accum_i8(mlp[0], mlp[1], report=False)
    ops.cov_accum_i8_multi([x, q, k], report=False)
def fused():
    ops.cov_accum_i8_multi([mlp, x, q, k], report=False)
for name, fn in (("sigma_mlp alone + three together", split), ("all four in one launch", fused), ("sigma_mlp alone + three together", split), ("all four in one launch", fused)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{kind:11s} {name:34s} {e0.elapsed_time(e1) / 6:7.3f} ms per batch")
