#!/usr/bin/env python3
"""bench.py -- layers compressed / second (covariance + decomposition + rebuild), Llama-3-8B @ 30 %.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one transformer layer taken through the whole hot path on synthetic Llama-3-8B-shaped inputs
(BASELINE.json configs[2]): 32 calibration batches of 16 x 2048 tokens (= 512 samples) pushed
through the four covariance hooks' kernels, sigma mirrored / normalised, then compress_nystrom + compress_qk +
compress_vo at keep ratio 0.7, compressed bf16 tensors and the rotary mask left resident in HBM.  Activations
and weights are resident in HBM before the timed region.  With N GPUs every rank compresses its own K layers
(weak scaling, no data-path collective) and ONE all-gather at the end of the timed region reassembles all N*K
compressed layers on every rank.

The JSON line also carries
  roofline     -- the dominant kernel.  Default covariance route (--cov-mode i8, exact int8 digit planes): i8_syrk_kernel on
                  sigma_mlp, int8 MFMA bound -- algorithmic int8 ops of its launches in the timed region (plane pairs x SYRK
                  count) / their summed durations, timed alone by HIP events the library records around that kernel on the
                  launch stream; the v_mfma_f64 kernel on the same batch is reported beside it as roofline.f64_route.
                  --cov-mode f64: cov_accum_multi_kernel, fp64 MFMA bound -- SYRK flops of the fused launches / their
                  durations (HIP events around each launch)
  cpu_baseline -- this repo's CPU oracle (torch-CPU fp64 restatement of the reference) timed on the host cores
                  on a bounded sample of the same workload (N = 1, rank 0 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from modegpt_amd import engine, ops, sharding  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X public fp64-matrix spec; bench also reports the measured issue rate
INT8_MFMA_PEAK_TOPS = 5000.0  # dense int8 = 2x the bf16 rate per clock (MI355X_MICROARCH.md, matrix cores table)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


class LaunchTimer:
    """HIP-event brackets around individual kernel launches on the current stream."""

    def __init__(self):
        self.pairs = []

    def run(self, flops, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.pairs.append((flops, e0, e1))

    def run_i8(self, ops_count, sigma, x):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()      # materialise the HIP events; the library re-records them around its product kernel
        e1.record()
        planes = ops.cov_accum_i8(sigma, x, events=(e0, e1))
        if planes:
            self.pairs.append((ops_count * planes * (planes + 1) // 30, e0, e1))   # ops_count is quoted for 5 planes = 15 pairs

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for _, a, b in self.pairs)
        fl = sum(f for f, _, _ in self.pairs)
        return len(self.pairs), fl, ms


def step(shape, adapter, layer_idx, batches, keep, n_texts, timer=None):
    """The whole hot path for one layer; returns its compressed tensors + rotary mask (resident in HBM)."""
    dev = batches[0]["h"].device
    covs = engine.new_covs(shape, dev)
    f, d, nh, nkv, hd = shape["d_ff"], shape["d"], shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]
    i8 = ops.COV_MODE == "i8" and shape["arch"] != "opt"
    for b in batches:
        t = b["h"].shape[0]
        if timer is None:
            engine.accumulate(covs, b, shape)
        elif not i8:  # the same fused launch as engine.accumulate (all four hooks of the layer), bracketed by events
            flops = t * (f * (f + 1) + d * (d + 1) + (nh + nkv) * hd * (hd + 1))   # SYRK count of the four problems
            timer.run(flops, lambda: engine.accumulate(covs, b, shape))
        else:  # engine.accumulate's i8 route spelled out, so that the dominant kernel (i8_syrk_kernel on sigma_mlp) can be
            #    timed alone: the library records the two events right around that launch
            timer.run_i8(15 * t * f * (f + 1), covs["mlp"], b["h"])     # 15 plane-pair products, SYRK count each
            ops.cov_accum_multi([(covs["x"], b["x"], 1), (covs["q"], b["q"], nh), (covs["k"], b["k"], nkv)], mode="i8")
    engine.finalize(covs, n_texts)
    tensors, mask = engine.compress_layer(adapter, layer_idx, covs, keep)
    return tensors, mask, covs


def cpu_baseline(shape, weights, covs_dev, sample_tokens, n_tokens_full, keep, ridges, gpu_out):
    """Oracle timed on the host: covariance on `sample_tokens` tokens (cost is exactly linear in tokens, scaled to
    the full count), decomposition + rebuild in full on the sigma of the GPU run (copied back), so the same call
    doubles as a full-size parity check of the GPU result."""
    from oracle import modegpt_oracle as O
    g = torch.Generator().manual_seed(7)
    f, d, nh, nkv, hd = shape["d_ff"], shape["d"], shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]
    acts = {k: torch.randn(sample_tokens, n, generator=g).to(torch.bfloat16)
            for k, n in (("h", f), ("x", d), ("q", nh * hd), ("k", nkv * hd))}
    sig = {"mlp": torch.zeros(f, f, dtype=torch.float64), "x": torch.zeros(d, d, dtype=torch.float64),
           "q": torch.zeros(nh, hd, hd, dtype=torch.float64), "k": torch.zeros(nkv, hd, hd, dtype=torch.float64)}
    t0 = time.perf_counter()
    O.cov_accum_tokens(sig["mlp"], acts["h"])
    O.cov_accum_tokens(sig["x"], acts["x"].view(1, sample_tokens, d))
    O.cov_accum_heads(sig["q"], acts["q"], nh, hd)
    O.cov_accum_heads(sig["k"], acts["k"], nkv, hd)
    t_cov_sample = time.perf_counter() - t0
    del sig, acts
    covs = {k: v.cpu() for k, v in covs_dev.items()}
    w = {k: v.cpu() for k, v in weights.items()}
    t0 = time.perf_counter()
    out = O.compress_layer_all(w, covs, shape, keep, ridges)
    t_dec = time.perf_counter() - t0
    t_cov_full = t_cov_sample * (n_tokens_full / sample_tokens)
    parity = {
        "mlp_idx_identical": bool(torch.equal(out["aux"]["mlp"][0], gpu_out["mlp_idx"].cpu())),
        "qk_mask_identical": bool(torch.equal(out["mask"], gpu_out["mask"].cpu())),
        "up_identical": bool(torch.equal(out["mlp"]["up"], gpu_out["up"].cpu())),
        "q_identical": bool(torch.equal(out["qk"]["q_proj"], gpu_out["q_proj"].cpu())),
        "down_max_rel": float(((out["mlp"]["down"].double() - gpu_out["down"].cpu().double()).abs().max()
                               / out["mlp"]["down"].double().abs().max()).item()),
    }
    return {
        "value": 1.0 / (t_cov_full + t_dec), "unit": "layers/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"oracle (torch-CPU fp64, {torch.get_num_threads()} threads): covariance of the 4 hooks timed on "
                   f"{sample_tokens} tokens = {t_cov_sample:.2f} s, scaled x{n_tokens_full // sample_tokens} to "
                   f"{n_tokens_full} tokens = {t_cov_full:.0f} s; mlp+qk+vo decomposition/rebuild of one layer timed "
                   f"in full = {t_dec:.2f} s"),
        "full_size_parity_vs_oracle": parity,
    }


def rope_gather_roofline(shape, keep, dev, launches=20):
    """SURVEY 8(f) row 3, measured beside the headline: the compressed model's rotary kernel (HBM bound) on one
    calibration-sized batch (16 x 2048 tokens) of q projections at this shape's compressed head width."""
    from modegpt_amd.compression.compress_qk import qk_rank_rule
    B, T, n_h, n_kv, hd = 16, 2048, shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]
    r = qk_rank_rule(hd, keep, shape["arch"])
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B, T, n_h * r, device=dev, generator=g).to(torch.bfloat16)
    ang = torch.rand(1, T, hd // 2, device=dev, generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos, sin = emb.cos().to(torch.bfloat16), emb.sin().to(torch.bfloat16)
    idx = torch.stack([torch.randperm(hd // 2, device=dev)[:r // 2] for _ in range(n_kv)])
    mask = torch.cat((idx, idx + hd // 2), dim=1)
    for _ in range(3):
        ops.rope_gather(x, cos, sin, mask, n_h, n_kv, hd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        ops.rope_gather(x, cos, sin, mask, n_h, n_kv, hd)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / launches * 1e3
    nbytes = 2 * x.numel() * x.element_size()            # every element read once and written once
    gbs = nbytes / (us * 1e-6) / 1e9
    return {"kernel": "rope_gather_kernel (compressed-head RoPE: cos/sin gathered by the rotary mask, fused transpose)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "avg_launch_us": us, "bytes_per_launch": nbytes, "launches": launches, "dtype": "bf16",
            "workload": f"q projection [16, 2048, {n_h} x {r}] of {hd}-wide heads, {n_kv} kv masks"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llama-3-8b", choices=sorted(engine.SHAPES))
    ap.add_argument("--batches", type=int, default=32, help="calibration batches per layer (32 x 16 = 512 samples)")
    ap.add_argument("--batch_size", type=int, default=16, help="samples of 2048 tokens per batch")
    ap.add_argument("--keep", type=float, default=0.7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cov-mode", default=None, choices=["f64", "i8"],
                    help="matrix cores for sigma_mlp / sigma_x: f64 (v_mfma_f64) or i8 (exact digit planes on v_mfma_i32_i8); "
                         "default: ops.COV_MODE (env MODEGPT_COV_MODE, else f64)")
    a = ap.parse_args()

    if a.cov_mode:
        ops.COV_MODE = a.cov_mode
    rank, world = sharding.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    shape = engine.SHAPES[a.model]
    tokens = a.batch_size * 2048
    n_texts = a.batches * a.batch_size

    batches = [engine.make_activation_batch(shape, tokens, seed=1234 * 1000 + b, device=dev) for b in range(a.batches)]
    n_layers_here = a.warmup + a.steps
    first = rank * n_layers_here
    layers = {first + i: engine.make_layer_weights(shape, 1234 + first + i, dev) for i in range(n_layers_here)}
    adapter = engine.TensorAdapter(shape, layers)
    ridges = dict(engine.RECIPE_RIDGES)

    for i in range(a.warmup):
        step(shape, adapter, first + i, batches, a.keep, n_texts)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timer = LaunchTimer()
    t0 = time.perf_counter()
    records, last = [], None
    for i in range(a.steps):
        li = first + a.warmup + i
        tensors, mask, covs = step(shape, adapter, li, batches, a.keep, n_texts, timer)
        records.append(sharding.pack_layer(li, {k: tensors.get(k) for k in sharding.TENSOR_ORDER}, mask))
        last = (li, tensors, mask, covs)
    gathered = sharding.allgather_records(records, a.steps, world)  # the single RCCL all-gather (no-op copy at N=1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert len(gathered) == world * a.steps

    n_launch, flops, ms = timer.summary()
    i8 = ops.COV_MODE == "i8" and shape["arch"] != "opt" and n_launch > 0
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch come from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of the same kernel on the
    # same four launch shapes (PMC passes cannot ride along a timed run); only valid for the default workload.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_cov_i8_hbm_traffic.json" if i8 else "r01_cov_hbm_traffic.json")
    if os.path.exists(tpath) and a.model == "llama-3-8b" and a.batch_size == 16:
        with open(tpath) as f:
            traffic = json.load(f)["hbm_bytes_per_launch"]
    out = {
        "metric": "transformer layers compressed/sec (covariance+decomp+rebuild), Llama-3-8B @30%",
        "value": world * a.steps / elapsed, "unit": "layers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8 digit planes of the bf16 activations (int32 accumulate, folded in f64; exact)" if i8 else "f64",
        "data": "synthetic",
        "config": {"workload": f"{a.model} shapes, {n_texts} calibration samples x 2048 tokens in {a.batches} batches "
                               f"of {a.batch_size}, keep ratio {a.keep} (compression {1 - a.keep:.0%}), ridges "
                               f"{ridges}, one layer per step per GPU", "layers_per_gpu": a.steps,
                   "parallelism": f"layer-sharded x{world}, one all-gather"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), "
                                     "profiles/r01_cov_hbm_traffic.json",
                     "kernel": "cov_accum_multi_kernel (v_mfma_f64_16x16x4_f64; sigma_mlp + sigma_x + sigma_q + sigma_k of "
                               "one calibration batch in one launch)", "launches": n_launch,
                     "avg_launch_ms": ms / n_launch, "flop_per_launch": flops / n_launch,
                     "flop_count": "SYRK: tokens * sum over the four problems of n * (n + 1) per launch"},
    }
    if i8:
        f = shape["d_ff"]
        # The five-plane kernel skips digit planes that are all-zero over a tile panel; `achieved` prices the MFMA work it
        # actually issued.  The executed / dense instruction ratio of the timed launches is read back from the library on a
        # replay of the same batches (the data is deterministic), outside the timed region.
        stats, routes_before = {}, dict(ops.I8_STATS)
        replay = torch.zeros(f, f, dtype=torch.float64, device=dev)
        for b in batches:
            ops.cov_accum_i8(replay, b["h"], mfma_stats=stats)
        del replay
        ops.I8_STATS.update(routes_before)
        executed_fraction = stats["executed"] / stats["dense"] if stats.get("dense") else 1.0
        dense_equivalent = achieved
        achieved = achieved * executed_fraction
        out["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s",
            "frac": achieved / INT8_MFMA_PEAK_TOPS, "executed_fraction": executed_fraction,
            "dense_equivalent_tops": dense_equivalent,
            "achieved_is": "int8 ops of the v_mfma instructions the kernel issued (mdg_cov_accum_i8_stats: executed / dense "
                           "instruction count of these batches x the dense op count below) / their summed durations; "
                           "dense_equivalent_tops counts the skipped all-zero planes as if multiplied",
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), profiles/r01_cov_i8_hbm_traffic.json",
            "kernel": "i8_syrk_kernel on sigma_mlp (v_mfma_i32_32x32x32_i8; the 15 (5 planes) or 21 (6 planes) digit-plane "
                      "pair products of one calibration batch per launch, timed alone by events the library records around it)",
            "launches": n_launch, "avg_launch_ms": ms / n_launch, "op_per_launch": flops / n_launch,
            "op_count": "dense count: plane pairs (15 or 21, see routes) x tokens x n (n + 1), the SYRK count of each product, "
                        "2 ops per multiply-add",
            "fp64_syrk_equivalent_tflops": n_launch * batches[0]["h"].shape[0] * f * (f + 1) / (ms * 1e-3) / 1e12,
            "routes": dict(ops.I8_STATS),
            "note": "sigma_x goes through the same kernel; sigma_q / sigma_k (1.4 % of the work) and any batch whose columns the "
                    "per-column depth statistic finds too heavy-tailed for six planes go through the v_mfma_f64 kernel (--cov-mode f64 runs everything there: "
                    "0.909 of the fp64 peak, DESIGN.md section 7)",
            "power_note": "while this kernel loops the device sits at its power cap (rocm-smi: 1330 W, sclk 1.94 GHz instead of 2.4; "
                          "scripts/probes/i8_clock_power.py): at that clock the int8 pipe peaks at 4.0 POP/s; `peak` above is the guide's 2.4 GHz figure"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        li, tensors, mask, covs = last
        gpu_out = dict(tensors)
        gpu_out["mask"] = mask
        # the selected index set is recoverable from the gathered `up` rows only indirectly; recompute it cheaply
        from modegpt_amd.compression.compress_mlp import _fl32
        sc = ops.ridge_scores(covs["mlp"], _fl32(ridges["nystrom_ridge"]))
        gpu_out["mlp_idx"] = ops.select_smallest_sorted(sc, int(shape["d_ff"] * a.keep))
        out["roofline"]["measured_mfma_f64_issue_rate_tflops"] = ops.probe_mfma_f64(4096)
        if i8:  # the same sigma_mlp batch through the v_mfma_f64 kernel, for the record
            scratch = torch.zeros_like(covs["mlp"])
            h = batches[0]["h"]
            ops.cov_accum(scratch, h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.cov_accum(scratch, h)
            ops.cov_accum(scratch, h)
            e1.record()
            torch.cuda.synchronize()
            tf = 2 * h.shape[0] * shape["d_ff"] * (shape["d_ff"] + 1) / (e0.elapsed_time(e1) * 1e-3) / 1e12
            out["roofline"]["f64_route"] = {"kernel": "cov_accum_kernel (v_mfma_f64_16x16x4_f64) on the same sigma_mlp batch",
                                            "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": tf / FP64_MFMA_PEAK_TFLOPS, "avg_launch_ms": e0.elapsed_time(e1) / 2}
            # ... and the six-plane route, which the depth statistic picks for the SiLU-gated MLP statistic of a real Llama
            # (the bench's Gaussian columns take five planes): a gated batch of the same shape through the same entry point
            if shape["arch"] != "opt":
                gen = torch.Generator(device=dev).manual_seed(4242)
                g = torch.randn(h.shape, generator=gen, device=dev, dtype=torch.float32)
                g = torch.nn.functional.silu(g).mul_(torch.randn(h.shape, generator=gen, device=dev, dtype=torch.float32))
                hg = g.to(torch.bfloat16)
                del g
                scratch.zero_()
                routes_before = dict(ops.I8_STATS)
                st6 = {}
                used = ops.cov_accum_i8(scratch, hg, events=(e0, e1), mfma_stats=st6)
                frac6 = st6["executed"] / st6["dense"] if st6.get("dense") else 1.0
                ms6 = []
                for _ in range(2):
                    used = ops.cov_accum_i8(scratch, hg, events=(e0, e1))
                    torch.cuda.synchronize()
                    ms6.append(e0.elapsed_time(e1))
                ops.I8_STATS.update(routes_before)      # not part of the timed region's route count
                pairs = {5: 15, 6: 21}.get(used)
                if pairs:
                    top = pairs * hg.shape[0] * shape["d_ff"] * (shape["d_ff"] + 1) / (sum(ms6) / len(ms6) * 1e-3) / 1e12
                    out["roofline"]["gated_route"] = {
                        "kernel": "i8_syrk_kernel on a SiLU-gated sigma_mlp batch of the same shape (outside the timed region)",
                        "planes": used, "achieved": top * frac6, "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                        "frac": top * frac6 / INT8_MFMA_PEAK_TOPS, "executed_fraction": frac6, "dense_equivalent_tops": top,
                        "avg_launch_ms": sum(ms6) / len(ms6)}
                del hg
            del scratch
        out["cpu_baseline"] = cpu_baseline(shape, layers[li], covs, 2048, n_texts * 2048, a.keep, ridges, gpu_out)
        if shape["arch"] != "opt":
            out["next_rows"] = {"rope_gather": rope_gather_roofline(shape, a.keep, dev)}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    sharding.finalize()


if __name__ == "__main__":
    main()
