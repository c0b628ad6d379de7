"""Does a low-priority side stream fill the tail of the sigma_mlp product with sigma_x's work?  Sequential vs overlapped
(side stream at default / lowest priority), wall time per batch pair and the sigma_mlp product's own duration by events."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from modegpt_amd import engine, ops

dev = torch.device("cuda:0")
shape = engine.SHAPES["llama-3-8b"]
b = engine.make_activation_batch(shape, 16 * 2048, seed=1, device=dev)
covs = engine.new_covs(shape, dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record()
nh, nkv = shape["n_heads"], shape["n_kv_heads"]


def seq():
    ops.cov_accum_i8(covs["mlp"], b["h"], events=(e0, e1))
    ops.cov_accum_multi([(covs["x"], b["x"], 1), (covs["q"], b["q"], nh), (covs["k"], b["k"], nkv)], mode="i8")


def make_overlapped(prio):
    side = torch.cuda.Stream(device=dev, priority=prio)

    def f():
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        ops.cov_accum_i8(covs["mlp"], b["h"], events=(e0, e1))
        with torch.cuda.stream(side):
            ops.cov_accum_multi([(covs["x"], b["x"], 1), (covs["q"], b["q"], nh), (covs["k"], b["k"], nkv)], mode="i8")
        main.wait_stream(side)
    return f


for label, fn in (("sequential", seq), ("overlapped, side priority 0", make_overlapped(0)),
                  ("overlapped, side priority lowest (%d)" % lo, make_overlapped(lo)), ("sequential", seq)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.time(); prod = 0.0
    for _ in range(16):
        fn()
        torch.cuda.synchronize()
        prod += e0.elapsed_time(e1)
    dt = (time.time() - t0) / 16 * 1e3
    print(f"{label}: {dt:.2f} ms per batch (all four statistics), sigma_mlp product {prod / 16:.2f} ms")
