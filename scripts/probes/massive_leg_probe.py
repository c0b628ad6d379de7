"""Where does bench.py's value_massive leg lose time?  One batch of the sigma_x + sigma_q + sigma_k fused launch, clean vs four
massive columns in x; and the decomposition chain on the resulting statistics (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modegpt_amd import engine, ops

dev = torch.device("cuda:0")
shape = engine.SHAPES["llama-3-8b"]
g = torch.Generator(device=dev).manual_seed(1)
b = engine.make_activation_batch(shape, 32768, seed=1234000, device=dev)
def massive(t, cols):
    t = t.clone()
    for i, c in enumerate(cols):
        top = t[:, c].float().abs().max()
        t[:, c] = (t[:, c].float() * 2.0 ** -(12 + i)).to(torch.bfloat16)
        t[torch.randperm(t.shape[0], device=dev, generator=g)[:3], c] = (top * 1.5).to(torch.bfloat16)
    return t
bm = dict(b); bm["x"] = massive(b["x"], (3, 1000, 2533, shape["d"] - 2)); bm["h"] = massive(b["h"], (5, 129, 7245, 14335))
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nh, nkv = shape["n_heads"], shape["n_kv_heads"]
for name, bb in (("clean", b), ("massive", bm)):
    covs = engine.new_covs(shape, dev)
    t_fused = timeit(lambda: ops.cov_accum_multi([(covs["x"], bb["x"], 1), (covs["q"], bb["q"], nh), (covs["k"], bb["k"], nkv)], mode="i8"))
    t_mlp = timeit(lambda: ops.cov_accum_i8(covs["mlp"], bb["h"], report=False))
    t_all = timeit(lambda: engine.accumulate(covs, bb, shape))
    info = []
    ops.cov_accum_i8_multi([(covs["x"], bb["x"], 1), (covs["q"], bb["q"], nh), (covs["k"], bb["k"], nkv)], report=True, route_info=info)
    print(f"{name}: sigma_x+q+k fused launch {t_fused:.3f} ms, sigma_mlp call {t_mlp:.3f} ms, all four hooks {t_all:.3f} ms; routes {[(i['planes'], i['columns']) for i in info]}")
    covs = engine.new_covs(shape, dev)
    for _ in range(4): engine.accumulate(covs, bb, shape)
    engine.finalize(covs, 64)
    w = engine.make_layer_weights(shape, 1234, dev)
    ad = engine.TensorAdapter(shape, {0: w})
    engine.compress_layer(ad, 0, covs, 0.7)
    t_dec = timeit(lambda: engine.compress_layer(ad, 0, covs, 0.7), n=2)
    from modegpt_amd.compression.compress_vo import compress_vo
    from modegpt_amd.compression.compress_mlp import compress_nystrom
    lst = lambda t: [t] + [None] * 31
    t_vo = timeit(lambda: compress_vo(adapter=ad, cov=lst(covs["x"]), keep_ratios=[0.7] * 32, target_layers=[0]), n=2)
    t_ml = timeit(lambda: compress_nystrom(adapter=ad, cov=lst(covs["mlp"]), keep_ratios=[0.7] * 32, target_layers=[0]), n=2)
    print(f"   decomposition chain {t_dec:.1f} ms (mlp {t_ml:.1f}, vo {t_vo:.1f})")
