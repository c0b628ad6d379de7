"""RCCL all_gather_into_tensor at the bench's real per-rank size (20 records x 340 MB = 6.8 GB > 2^31 elements of uint8) on ONE
rank: does torch / RCCL take the count, and what does the single-rank path cost?  (the 8-rank run is the driver's)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from modegpt_amd import sharding
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 170_000_000
recs = [torch.full((n + 7 * i,), i + 1, dtype=torch.int16, device="cuda") for i in range(20)]
torch.cuda.synchronize(); t0 = time.time()
out = sharding.allgather_records(recs, 20, 1, force_collective=True)
torch.cuda.synchronize(); t = time.time() - t0
ok = all(int(o[0]) == i + 1 and int(o[recs[i].numel() - 1]) == i + 1 for i, o in enumerate(out))
print(f"all-gather of {sum(r.numel() for r in recs) * 2 / 1e9:.2f} GB on one rank: {t:.3f} s, {len(out)} records, contents {'ok' if ok else 'WRONG'}")
dist.destroy_process_group()
