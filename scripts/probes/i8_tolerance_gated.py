"""The int8 route's tolerance dial (ops.set_i8_tolerance) on SiLU-gated sigma_mlp data at the bench's full size: route, bound, measured
error of one batch against the fp64 kernel, step time, and the compressed tensors of a few layers against the fp64 route's.
    python scripts/probes/i8_tolerance_gated.py [factor=64] [layers=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from modegpt_amd import engine, ops

factor = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
n_layers = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
shape = engine.SHAPES["llama-3-8b"]
tokens, n_batches = 16 * 2048, 32
n_texts = n_batches * 16
batches = [engine.make_activation_batch(shape, tokens, seed=1234 * 1000 + b, device=dev) for b in range(n_batches)]
gated = []
for b, bt in enumerate(batches):
    gen = torch.Generator(device=dev).manual_seed(4242 + b)
    g = torch.nn.functional.silu(torch.randn(bt["h"].shape, generator=gen, device=dev, dtype=torch.float32))
    g.mul_(torch.randn(bt["h"].shape, generator=gen, device=dev, dtype=torch.float32))
    gated.append({"h": g.to(torch.bfloat16), "x": bt["x"], "q": bt["q"], "k": bt["k"]})
    del g
del batches
ids = list(range(n_layers))
adapter = engine.TensorAdapter(shape, {i: engine.make_layer_weights(shape, 1234 + i, dev) for i in ids})
f = shape["d_ff"]
S64 = torch.zeros(f, f, dtype=torch.float64, device=dev)
ops.cov_accum(S64, gated[0]["h"])
for fac in (1.0, factor):
    ops.set_i8_tolerance(fac)
    S8 = torch.zeros(f, f, dtype=torch.float64, device=dev)
    info, st = {}, {}
    used = ops.cov_accum_i8(S8, gated[0]["h"], mfma_stats=st, route_info=info)
    print(f"tolerance x{fac:g}: one batch of sigma_mlp -> {used} planes, bound {info['bound']:.3e} (columns {info['columns']}), measured "
          f"{bench.entrywise_err(S8, S64):.3e} of sqrt(s_ii s_jj) against the fp64 kernel, executed / dense MFMAs {st['executed'] / st['dense']:.3f}")
    del S8
del S64
outs = {}
for name, mode, fac in (("f64", "f64", 1.0), ("i8 x1", "i8", 1.0), (f"i8 x{factor:g}", "i8", factor)):
    ops.COV_MODE = mode
    ops.set_i8_tolerance(fac)
    bench.step(shape, adapter, ids[0], gated, 0.7, n_texts)
    timer = bench.LaunchTimer()
    sec, res = bench.timed_steps(shape, adapter, ids, gated, 0.7, n_texts, timer, True)
    nl, _, ms = timer.summary()
    outs[name] = res
    print(f"{name}: {sec / len(ids) * 1e3:.1f} ms per layer = {len(ids) / sec:.4f} layers/s" + (f", sigma_mlp product launch {ms / nl:.2f} ms" if nl else ""))
ops.set_i8_tolerance(1.0)
for name in list(outs)[1:]:
    head = {li: (t, m) for li, t, m, _ in outs[name]}
    print(name, "against the fp64 route:", bench.compare_outputs(head, outs["f64"], n_texts * 2048))
