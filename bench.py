#!/usr/bin/env python3
"""bench.py -- layers compressed / second (covariance + decomposition + rebuild), Llama-3-8B @ 30 %.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under torch.distributed.run -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment -- or plainly as
     above: with WORLD_SIZE unset the script starts its own N ranks as fresh child processes, one per GPU, BEFORE it touches the
     GPU, waits for them and exits with the worst of their exit codes)

One STEP = one transformer layer taken through the whole hot path on synthetic Llama-3-8B-shaped inputs
(BASELINE.json configs[2]): 32 calibration batches of 16 x 2048 tokens (= 512 samples) pushed
through the four covariance hooks' kernels, sigma mirrored / normalised, then compress_nystrom + compress_qk +
compress_vo at keep ratio 0.7, compressed bf16 tensors and the rotary mask left resident in HBM.  Activations
and weights are resident in HBM before the timed region.  With N GPUs every rank compresses its own K layers
(weak scaling, no data-path collective) and ONE all-gather at the end of the timed region reassembles all N*K
compressed layers on every rank.

The JSON line carries numbers and short labels only ("The bench line, field by field", explains each):
  roofline     -- the dominant kernel.  Default covariance route (--cov-mode i8: error-free split into int8 digit planes, truncated
                  plane-pair product): i8_syrk_kernel on sigma_mlp, int8 MFMA bound, priced on the MFMAs it ISSUED, timed alone by
                  HIP events the library records around it on the launch stream; .algorithmic restates it in SURVEY 8(d)'s fp64-SYRK
                  units; .error_bound is the per-call guaranteed bound the device computed; .f64_route the v_mfma_f64 kernel on the
                  same batch.  --cov-mode f64: cov_accum_multi_kernel, fp64 MFMA bound
  cpu_baseline -- this repo's CPU oracle (torch-CPU fp64 restatement of the reference) timed on the host cores on a bounded
                  sample of the same workload (N = 1, rank 0 only); full_size_parity_vs_oracle: its outputs against the GPU's
  value_f64_route / value_gated / value_massive -- the SAME step loop timed again (N = 1 only, after the headline): everything
                  on v_mfma_f64; SiLU-gated sigma_mlp activations (what a real Llama MLP feeds the hook: six planes); four
                  massive-activation columns in sigma_mlp and sigma_x (they leave the launch alone, through the fp64 column kernel)
  i8_vs_f64_outputs -- the headline leg's compressed tensors against the f64 leg's, same layers, full token count
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from modegpt_amd import engine, ops, sharding  # noqa: E402  (imports only: nothing here touches the GPU)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X public fp64-matrix spec; bench also reports the measured issue rate
INT8_MFMA_PEAK_TOPS = 5000.0  # dense int8 = 2x the bf16 rate per clock (MI355X_MICROARCH.md, matrix cores table)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


class LaunchTimer:
    """HIP-event brackets around individual kernel launches on the current stream."""

    def __init__(self):
        self.pairs = []
        self.dec = []

    def run(self, flops, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.pairs.append((flops, e0, e1))

    def run_i8(self, syrk_count, sigma, x):
        """Enqueue only (the route is chosen on the device, nothing here waits for the host); the launch is priced afterwards,
        once the replay outside the timed region has told how many digit planes this batch takes: price_i8()."""
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()      # materialise the HIP events; the library re-records them around its product launches
        e1.record()
        ops.cov_accum_i8(sigma, x, events=(e0, e1), report=False)
        self.pairs.append((syrk_count, e0, e1))

    def price_i8(self, planes_per_batch):
        """SYRK count -> int8 ops: x plane pairs (15 for five planes, 21 for six) of the batch each launch worked on (launches
        cycle through the batches in order); launches whose batch fell back to the fp64 kernel are dropped."""
        nb, priced = len(planes_per_batch), []
        for i, (cnt, e0, e1) in enumerate(self.pairs):
            p = planes_per_batch[i % nb]
            if p:
                priced.append((cnt * {3: 9, 5: 15, 6: 21}[p], e0, e1))
        self.pairs = priced

    def run_decomposition(self, flops, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.dec.append((flops, e0, e1))
        return out

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for _, a, b in self.pairs)
        fl = sum(f for f, _, _ in self.pairs)
        return len(self.pairs), fl, ms

    def decomposition_summary(self):
        torch.cuda.synchronize()
        return len(self.dec), sum(f for f, _, _ in self.dec), sum(a.elapsed_time(b) for _, a, b in self.dec)


def decomposition_flops(shape, keep):
    """fp64 flops the compress_* chain EXECUTES for one layer (this engine's formulation, DESIGN.md sections 3 and 5), big
    terms only: ridge scores = Cholesky n^3/3 + triangular inverse n^3/3 (compress_mlp.py:13-25); Nystrom = gathered product
    C[idx,:] W_d^T 2 r n d + Cholesky of C_kk r^3/3 + two triangular solves 2 r^2 d (compress_mlp.py:52-62); VO through the
    Gram route = W_v C 2 (n_kv hd) d^2 + per-head Grams and factor products (compress_vo.py:112-223).  QK is a selection."""
    n, d, nh, nkv, hd = shape["d_ff"], shape["d"], shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]
    r = int(n * keep)
    from modegpt_amd.compression.compress_qk import qk_rank_rule
    rv = qk_rank_rule(hd, keep, shape["arch"])
    ridge = 2.0 * n ** 3 / 3.0
    nystrom = 2.0 * r * n * d + r ** 3 / 3.0 + 2.0 * r * r * d
    vo = 2.0 * nkv * hd * d * d + nkv * 2.0 * hd * d * hd + nkv * 2.0 * rv * hd * d + nh * 2.0 * d * hd * rv
    return ridge + nystrom + vo


def accumulate_layer(shape, batches, n_texts, timer=None):
    """First half of a step: the four covariance hooks' kernels over all calibration batches + mirror / normalise.  Only enqueues
    (the int8 route picks its path on the device), so the host is free again at once."""
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
            timer.run_i8(t * f * (f + 1), covs["mlp"], b["h"])          # SYRK count; x 15 or 21 plane-pair products afterwards
            ops.cov_accum_multi([(covs["x"], b["x"], 1), (covs["q"], b["q"], nh), (covs["k"], b["k"], nkv)], mode="i8")
    engine.finalize(covs, n_texts)
    return covs


def compress_layer(shape, adapter, layer_idx, covs, keep, timer=None):
    """Second half of a step: compress_nystrom + compress_qk + compress_vo on the layer's finished statistics."""
    if timer is None:
        return engine.compress_layer(adapter, layer_idx, covs, keep, check=False)
    return timer.run_decomposition(decomposition_flops(shape, keep), lambda: engine.compress_layer(adapter, layer_idx, covs, keep, check=False))


def step(shape, adapter, layer_idx, batches, keep, n_texts, timer=None):
    """The whole hot path for one layer; returns its compressed tensors + rotary mask (resident in HBM)."""
    covs = accumulate_layer(shape, batches, n_texts, timer)
    tensors, mask = compress_layer(shape, adapter, layer_idx, covs, keep, timer)
    return tensors, mask, covs


class Pipeline:
    """Steps back to back with the decomposition chains of the previous `depth` layers on side streams (one each)
    while the next layer's covariance kernels run on the caller's stream.  A chain is latency-bound (191 dependent
    single-workgroup Cholesky steps between small GEMMs) and leaves most of the chip idle between its large GEMMs; the covariance
    of the next layer does not depend on it, and neither does another layer's chain: two chains side by side fill each other's
    gaps.  Same kernels, same inputs, same results -- only the order in which the streams' workgroups reach the CUs changes.
    Each layer's statistics live in their own buffers until its chain is done."""

    def __init__(self, shape, adapter, batches, keep, n_texts, timer=None, enabled=True, depth=2):
        self.a = (shape, adapter, batches, keep, n_texts, timer)
        self.enabled = enabled
        self.depth = depth if enabled else 1
        dev = batches[0]["h"].device
        self.main = torch.cuda.current_stream(dev)
        # (same priority as the caller's stream: with high-priority side streams every short chain kernel overtakes the covariance
        #  launch that is ready next, and the loop is 1.5 % slower -- 962 / 956 against 946 / 944 ms per step on one box)
        self.sides = [torch.cuda.Stream(device=dev) for _ in range(self.depth)] if enabled else []
        self.pending = []
        self.done = []          # (layer, tensors, mask, covs) in layer order

    def _finish(self):
        shape, adapter, _, keep, _, timer = self.a
        for slot, (li, covs, ev) in enumerate(self.pending):
            if not self.enabled:
                tensors, mask = compress_layer(shape, adapter, li, covs, keep, timer)
            else:
                side = self.sides[slot]
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    tensors, mask = compress_layer(shape, adapter, li, covs, keep, timer)
                for t in covs.values():
                    t.record_stream(side)       # allocated on the main stream, last read on the side stream
            if self.done:               # (only the latest layer's statistics are looked at afterwards: 1.9 GB per layer otherwise)
                pl, pt, pm, _ = self.done[-1]
                self.done[-1] = (pl, pt, pm, None)
            self.done.append((li, tensors, mask, covs))
        self.pending = []

    def submit(self, layer_idx, last=False):
        shape, _, batches, _, n_texts, timer = self.a
        covs = accumulate_layer(shape, batches, n_texts, timer)      # enqueued on the main stream (the host runs ~2 layers ahead)
        ev = torch.cuda.Event()
        ev.record(self.main)
        if len(self.pending) == self.depth or (last and self.pending):
            self._finish()                                            # the previous layers' chains, beside this covariance
        self.pending.append((layer_idx, covs, ev))                    # (last: no covariance follows -- nothing is held back for it)

    def drain(self):
        if self.pending:
            self._finish()
        for side in self.sides:
            self.main.wait_stream(side)         # whatever follows on the caller's stream sees the compressed tensors
        return self.done


def entrywise_err(S, R):
    """max over the lower triangle of |S - R|_ij / sqrt(R_ii R_jj) (row chunks: the temporaries of a 14336^2 matrix are 1.6 GB)."""
    n = R.shape[-1]
    d = torch.sqrt(torch.diagonal(R, dim1=-2, dim2=-1))
    d = torch.where(d > 0, d, torch.ones_like(d))
    worst = 0.0
    for r0 in range(0, n, 2048):
        r1 = min(n, r0 + 2048)
        e = (S[..., r0:r1, :] - R[..., r0:r1, :]).abs() / (d[..., r0:r1, None] * d[..., None, :])
        rows = torch.arange(r0, r1, device=R.device)[:, None]
        cols = torch.arange(n, device=R.device)[None, :]
        worst = max(worst, torch.where(cols <= rows, e, torch.zeros_like(e)).max().item())
    return worst


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(shape, layer_cases, sample, n_tokens_full, keep, ridges, gpu_sig_sample):
    """Oracle timed on the host (SURVEY 8d / BASELINE.md section 3): covariance of the four hooks on `sample` -- ONE FULL calibration
    batch (16 x 2048 tokens of batch 0, copied to the host; the cost is exactly linear in tokens and is scaled to the full count.
    BASELINE.md asks for 8 batches; one is ~30 s on 128 threads and eight would take the default bench run past the few minutes the
    contract allows, so the sample is one batch and says so) -- and decomposition + rebuild IN FULL for every entry of
    `layer_cases` = [(label, weights, sigma (device), gpu outputs)], two layers: the last timed layer and one layer of the
    distinct-activation set.  The same calls double as full-size parity checks of the GPU results: the oracle's sigma of the sample
    against what the GPU's default route (gpu_sig_sample) made of the same rows, entry-wise, and the oracle's compressed tensors
    against the GPU's for both layers."""
    from oracle import modegpt_oracle as O
    f, d, nh, nkv, hd = shape["d_ff"], shape["d"], shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]
    sample_tokens = sample["h"].shape[0]
    sig = {"mlp": torch.zeros(f, f, dtype=torch.float64), "x": torch.zeros(d, d, dtype=torch.float64),
           "q": torch.zeros(nh, hd, hd, dtype=torch.float64), "k": torch.zeros(nkv, hd, hd, dtype=torch.float64)}
    t0 = time.perf_counter()
    if shape["arch"] == "opt":
        O.cov_accum_tokens_relu(sig["mlp"], sample["h"])
    else:
        O.cov_accum_tokens(sig["mlp"], sample["h"])
    O.cov_accum_tokens(sig["x"], sample["x"].view(1, sample_tokens, d))
    O.cov_accum_heads(sig["q"], sample["q"], nh, hd)
    O.cov_accum_heads(sig["k"], sample["k"], nkv, hd)
    t_cov_sample = time.perf_counter() - t0
    sigma_err = {k: entrywise_err(gpu_sig_sample[k].cpu(), sig[k]) for k in sig}
    del sig
    t_decs, parity = [], {"sigma_vs_oracle_entrywise_max": sigma_err, "sigma_tokens": sample_tokens, "layers": []}
    for label, weights, covs_dev, gpu_out in layer_cases:
        covs = {k: v.cpu() for k, v in covs_dev.items()}
        w = {k: v.cpu() for k, v in weights.items()}
        t0 = time.perf_counter()
        out = O.compress_layer_all(w, covs, shape, keep, ridges)
        t_decs.append(time.perf_counter() - t0)
        parity["layers"].append({
            "layer": label,
            "mlp_idx_identical": bool(torch.equal(out["aux"]["mlp"][0], gpu_out["mlp_idx"].cpu())),
            "qk_mask_identical": bool(torch.equal(out["mask"], gpu_out["mask"].cpu())),
            "up_identical": bool(torch.equal(out["mlp"]["up"], gpu_out["up"].cpu())),
            "q_identical": bool(torch.equal(out["qk"]["q_proj"], gpu_out["q_proj"].cpu())),
            "down_max_rel": float(((out["mlp"]["down"].double() - gpu_out["down"].cpu().double()).abs().max()
                                   / out["mlp"]["down"].double().abs().max()).item())})
        del covs, w, out
    t_dec = sum(t_decs) / len(t_decs)
    t_cov_full = t_cov_sample * (n_tokens_full / sample_tokens)
    return {
        "value": 1.0 / (t_cov_full + t_dec), "unit": "layers/s", "cores": torch.get_num_threads(), "nproc": os.cpu_count(),
        "cpu": cpu_model_name(), "kind": "port",
        "measured_s": {"covariance_sample": t_cov_sample, "decomposition_per_layer": t_decs},
        "sample": (f"oracle, torch-CPU fp64: covariance of one full batch ({sample_tokens} tokens) x{n_tokens_full // sample_tokens} "
                   f"= {t_cov_full:.0f} s extrapolated (linear); decomposition + rebuild of {len(t_decs)} layers in full"),
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
    return {"kernel": "rope_gather_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "avg_launch_us": us, "workload": f"q [16, 2048, {n_h} x {r}] bf16"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llama-3-8b", choices=sorted(engine.SHAPES))
    ap.add_argument("--batches", type=int, default=32, help="calibration batches per layer (32 x 16 = 512 samples)")
    ap.add_argument("--batch_size", type=int, default=16, help="samples of 2048 tokens per batch")
    ap.add_argument("--keep", type=float, default=0.7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", dest="pipeline", action="store_true", default=True,
                    help="(default) layer L's decomposition chain runs on a side stream beside layer L + 1's covariance kernels: "
                         "the chain enqueues without a host round trip (deferred status), so its kernels fill what the covariance "
                         "launches leave idle; +3 %% on the step (A/B/A/B on one box, DESIGN.md section 7)")
    ap.add_argument("--chain-depth", type=int, default=2,
                    help="how many layers' decomposition chains run side by side (each on a stream of its own) beside the next covariance")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="one after the other on one stream: per-kernel figures without interference (reported as value_sequential)")
    ap.add_argument("--extra-leg-steps", type=int, default=8,
                    help="layers per extra leg (value_f64_route is capped at 6: its launches are 4.5x longer), min'ed with --steps: the extra "
                         "legs are measurements beside the headline and must not turn the default run into a quarter of an hour")
    ap.add_argument("--distinct-layers", type=int, default=3,
                    help="layers of the i8-vs-f64 output comparison whose activations are generated per (layer, batch) -- distinct "
                         "sigma per layer (SURVEY 8d: seed (1234, layer, batch)); outside every timed region")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip value_f64_route / value_gated (the same step loop on the fp64 route and on SiLU-gated data)")
    ap.add_argument("--cov-mode", default=None, choices=["f64", "i8"],
                    help="matrix cores for sigma_mlp / sigma_x: i8 (fp64 result emulated by digit planes on v_mfma_i32_i8, <= 1e-12; "
                         "the default, ops.COV_MODE / env MODEGPT_COV_MODE) or f64 (v_mfma_f64, the reference's accumulation)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no kernels: launcher, rendezvous, record packing and the one all-gather on synthetic records "
                         "(runs on CPU / gloo; prints value null)")
    return ap.parse_args(argv)


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start N ranks of this same command line as FRESH child
    processes (one per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), wait,
    and return the worst exit code.  Called before this process has made any GPU call: the parent never initialises HIP, the
    children start from a clean interpreter (no fork of a GPU context, no exec from an initialised one)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, pending = 0, set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                for o in pending:            # one rank failed: the others would sit in a collective until its timeout
                    procs[o].terminate()
        time.sleep(0.2)
    return worst


def plumbing_only(a, rank, world):
    """Everything around the kernels on synthetic records: rendezvous, barrier, pack_layer, the padded all-gather, unpack."""
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if torch.cuda.is_available() else torch.device("cpu")
    g = torch.Generator().manual_seed(100 + rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    records = []
    for i in range(a.steps):
        li, r = rank * a.steps + i, 8 + rank + i            # ragged: ranks differ per layer
        t = {k: torch.randn(r, 16, generator=g).to(torch.bfloat16).to(dev) for k in ("up", "gate", "q_proj", "k_proj", "v_proj")}
        t["down"] = torch.randn(16, r, generator=g).to(torch.bfloat16).to(dev)
        t["o_proj"] = torch.randn(16, r, generator=g).to(torch.bfloat16).to(dev)
        records.append(sharding.pack_layer(li, t, torch.arange(r).reshape(1, r) + li))
    gathered = sharding.allgather_records(records, a.steps, world)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    layers = sorted(sharding.unpack_layer(rec)[0] for rec in gathered)
    assert layers == list(range(world * a.steps)), layers
    if rank == 0:
        print(json.dumps({"metric": "transformer layers compressed/sec (covariance+decomp+rebuild), Llama-3-8B @30%", "value": None,
                          "unit": "layers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "plumbing_only": True,
                          "backend": dist.get_backend() if world > 1 else None, "gathered_layers": len(layers),
                          "allgather_s": elapsed}))
    sharding.finalize()


def timed_steps(shape, adapter, layer_ids, batches, keep, n_texts, timer=None, pipelined=True, depth=2):
    """K steps back to back between two synchronisations; returns (seconds, per-step outputs)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe = Pipeline(shape, adapter, batches, keep, n_texts, timer, enabled=pipelined, depth=depth)
    for li in layer_ids:
        pipe.submit(li, last=li == layer_ids[-1])
    outs = pipe.drain()
    adapter.check_chains()          # the layers' Cholesky / eigensolver statuses, read once for all of them
    torch.cuda.synchronize()
    return time.perf_counter() - t0, outs


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    if a.cov_mode:
        ops.COV_MODE = a.cov_mode
    rank, world = sharding.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.plumbing_only:
        return plumbing_only(a, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py measures the HIP path and there is no GPU here (no CPU fallback exists by design); "
                         "--plumbing-only exercises the launcher and the all-gather without kernels")
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
    adapter.calib_tokens = n_texts * 2048      # (what the selection certificate's fp64 rounding bound scales with)
    ridges = dict(engine.RECIPE_RIDGES)
    tol = ops.i8_tolerance()                    # the int8 route's tolerance factor of the headline (1 unless MODEGPT_I8_TOLERANCE says otherwise)

    warm, gather_buffers = None, None
    for i in range(a.warmup):
        warm = step(shape, adapter, first + i, batches, a.keep, n_texts)
    adapter.check_chains()
    adapter.report_selection_margins()          # (the warm-up layers' certificates: not part of the timed layers' summary)
    if warm is not None:
        # the gather's one-time costs belong to the warm-up too: RCCL sets up its all-gather channels on the first call, and the
        # send / receive buffers of the timed gather (K records per rank, ~0.34 GB each, x world on the receiving side: 54 GB at
        # 8 ranks) come out of the caching allocator instead of a fresh hipMalloc inside the timed region
        rec = sharding.pack_layer(first + a.warmup - 1, {k: warm[0].get(k) for k in sharding.TENSOR_ORDER}, warm[1])
        sharding.allgather_records([rec], 1, world)
        gather_buffers = sharding.gather_buffers(a.steps, world, rec.numel() + 64, dev)    # (every layer: same shapes, same keep ratio)
        del rec, warm
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timer = LaunchTimer()
    pipelined = a.pipeline
    t0 = time.perf_counter()
    pipe = Pipeline(shape, adapter, batches, a.keep, n_texts, timer, enabled=pipelined, depth=a.chain_depth)
    for i in range(a.steps):
        pipe.submit(first + a.warmup + i, last=i == a.steps - 1)
    records, last, done = [], None, []
    for li, tensors, mask, covs in pipe.drain():
        records.append(sharding.pack_layer(li, {k: tensors.get(k) for k in sharding.TENSOR_ORDER}, mask))
        last = (li, tensors, mask, covs)
        done.append((li, tensors, mask, None))          # (compressed tensors of every timed layer: compared with the f64 leg's below)
    adapter.check_chains()                               # the layers' Cholesky / eigensolver statuses, read once for all of them
    selection = adapter.report_selection_margins()        # (64 bytes per layer, behind the same wait: certificates of the MLP rank selections)
    gathered = sharding.allgather_records(records, a.steps, world, buffers=gather_buffers)  # the single RCCL all-gather (no-op copy at N=1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert len(gathered) == world * a.steps
    del gathered, records

    i8 = ops.COV_MODE == "i8" and shape["arch"] != "opt"
    stats, bounds = {}, []
    if i8:
        # The product kernel skips digit planes that are all-zero over a tile panel; `achieved` prices the MFMA work it
        # actually issued.  Route, error bound and executed / dense instruction ratio of the timed launches are read back from the
        # library on a replay of the same batches (the data and the integer route statistics are deterministic), outside the timed region.
        replay = torch.zeros(shape["d_ff"], shape["d_ff"], dtype=torch.float64, device=dev)
        routes_before = dict(ops.I8_STATS)
        planes_per_batch = []
        for b in batches:
            info = {}
            cls = ops.cov_accum_i8(replay, b["h"], mfma_stats=stats, route_info=info)
            planes_per_batch.append(3 if (cls and info["exact"]) else cls)       # (the product kernel that ran: 3 = the exact route's nine-pair launch)
            bounds.append(info)
        ops.I8_STATS.update(routes_before)
        del replay
        timer.price_i8(planes_per_batch)
    n_launch, flops, ms = timer.summary()
    n_dec, dec_flops, dec_ms = timer.decomposition_summary()
    i8 = i8 and n_launch > 0
    achieved = flops / (ms * 1e-3) / 1e12 if n_launch else 0.0
    # HBM bytes per launch come from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of the same kernel on the
    # same launch shape (PMC passes cannot ride along a timed run); only valid for the default workload.
    traffic, tfile = None, None
    for cand in (("r04_cov_i8_hbm_traffic.json", "r03_cov_i8_hbm_traffic.json") if i8 else ("r01_cov_hbm_traffic.json",)):
        tpath = os.path.join(ROOT, "profiles", cand)
        if os.path.exists(tpath) and a.model == "llama-3-8b" and a.batch_size == 16:
            with open(tpath) as f:
                traffic, tfile = json.load(f)["hbm_bytes_per_launch"], cand
            break
    dec_tf = dec_flops / (dec_ms * 1e-3) / 1e12 if n_dec else None
    decomposition = None if not n_dec else {
        "bound": "mfma", "achieved": dec_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dec_tf / FP64_MFMA_PEAK_TFLOPS,
        "avg_ms_per_layer": dec_ms / n_dec, "flop_per_layer": dec_flops / n_dec,
        "timed": "on the side stream beside the next layer's covariance (waits for CUs included); alone: value_sequential.decomposition"
                 if pipelined else "alone on the caller's stream"}
    # The JSON line carries numbers and short labels only; what each field means is written down in DESIGN.md section 7
    # ("The bench line, field by field").
    out = {
        "metric": "transformer layers compressed/sec (covariance+decomp+rebuild), Llama-3-8B @30%",
        "value": world * a.steps / elapsed, "unit": "layers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": (f"f64 emulated on int8 MFMA digit planes (error <= {1.1e-11 * tol:.3g} guaranteed per call, typically 1e-13)"
                  if i8 else "f64"),
        "data": "synthetic",
        "config": {"workload": f"{a.model} shapes, {n_texts} samples x 2048 tokens in {a.batches} batches of {a.batch_size}, keep {a.keep}, "
                               f"tests.sh ridges, Gaussian columns x log-uniform[0.05, 2] scales (SURVEY 8d)",
                   "layers_per_gpu": a.steps, "parallelism": f"layer-sharded x{world}, one all-gather", "pipelined": pipelined},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_from": tfile,
                     "kernel": "cov_accum_multi_kernel (v_mfma_f64_16x16x4_f64)", "launches": n_launch,
                     "avg_launch_ms": ms / n_launch, "flop_per_launch": flops / n_launch},
    }
    if i8:
        f = shape["d_ff"]
        executed_fraction = stats["executed"] / stats["dense"] if stats.get("dense") else 1.0
        dense_equivalent = achieved
        achieved = achieved * executed_fraction
        syrk_tf = n_launch * batches[0]["h"].shape[0] * f * (f + 1) / (ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s",
            "frac": achieved / INT8_MFMA_PEAK_TOPS, "traffic": traffic, "traffic_from": tfile,
            "kernel": "i8_syrk_kernel on sigma_mlp (v_mfma_i32_32x32x32_i8), issued MFMAs", "launches": n_launch,
            "avg_launch_ms": ms / n_launch, "op_per_launch": flops / n_launch, "executed_fraction": executed_fraction,
            "dense_equivalent_tops": dense_equivalent,
            "algorithmic": {"fp64_syrk_tflops": syrk_tf, "x_fp64_mfma_peak": syrk_tf / FP64_MFMA_PEAK_TFLOPS,
                            "frac_of_int8_peak": syrk_tf / INT8_MFMA_PEAK_TOPS},
            "routes": ops.i8_route_counts(dev),
            "error_bound": {"guaranteed_max": max(b_["bound"] for b_ in bounds), "sq_max": max(b_["sq"] for b_ in bounds),
                            "x_max": max(b_["x"] for b_ in bounds), "planes": sorted(set(planes_per_batch)),
                            "exact_route_batches": sum(1 for b_ in bounds if b_["exact"]),
                            "fp64_columns_per_batch_max": max(len(b_["columns"]) for b_ in bounds)}}
    if i8:
        out["roofline"]["i8_tolerance_factor"] = tol
        if bounds and all(b_["exact"] for b_ in bounds):     # the headline's sigma_mlp calls ran the exact route: no plane pair dropped
            out["dtype"] = ("f64 from int8 MFMA digit planes, exact route: nine plane pairs + fp64 remainder sums, nothing truncated "
                            f"(error <= {max(b_['bound'] for b_ in bounds):.1g} per call: fp64 rounding)")
    if selection:
        out["selection_certificate"] = summarise_selection(selection)
    if not pipelined:          # (beside the covariance its events also time the waits for CUs: reported from the sequential leg below instead)
        out["roofline"]["decomposition"] = decomposition
    ids = [first + a.warmup + i for i in range(a.steps)]
    headline_out = {li: (tensors, mask) for li, tensors, mask, _ in done}
    # i8 against f64 on layers with DISTINCT statistics (seed (1234, layer, batch) and per-layer column scales; generated one batch
    # at a time, outside every timed region): compressed tensors of the two routes, their selection certificates
    distinct, kept = None, None
    if rank == 0 and world == 1 and i8 and a.distinct_layers > 0 and not (a.no_cpu_baseline and a.no_extra_legs):
        distinct, kept = distinct_sigma_parity(shape, adapter, ids[:a.distinct_layers], a, dev, n_texts, tokens)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        li, tensors, mask, covs = last
        gpu_out = dict(tensors)
        gpu_out["mask"] = mask
        # the selected index set is recoverable from the gathered `up` rows only indirectly; recompute it cheaply
        from modegpt_amd.compression.compress_mlp import _fl32
        sc = ops.ridge_scores(covs["mlp"], _fl32(ridges["nystrom_ridge"]))
        gpu_out["mlp_idx"] = ops.select_smallest_sorted(sc, int(shape["d_ff"] * a.keep))
        out["roofline"]["measured_mfma_f64_issue_rate_tflops"] = ops.probe_mfma_f64(4096)
        if i8:
            zero, rnd = ops.probe_mfma_i8(random_operands=False), ops.probe_mfma_i8(random_operands=True)
            out["roofline"]["measured_i8_mfma_rate_tops"] = {"zero_operands": zero, "random_operands": rnd,
                                                             "frac_of_random_operand_rate": out["roofline"]["achieved"] / rnd}
        h = batches[0]["h"]
        if i8:  # the same sigma_mlp batch through the v_mfma_f64 kernel, for the record, and the two routes against each other
            scratch = torch.zeros_like(covs["mlp"])
            ops.cov_accum(scratch, h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.cov_accum(scratch, h)
            ops.cov_accum(scratch, h)
            e1.record()
            torch.cuda.synchronize()
            tf = 2 * h.shape[0] * shape["d_ff"] * (shape["d_ff"] + 1) / (e0.elapsed_time(e1) * 1e-3) / 1e12
            out["roofline"]["f64_route"] = {"kernel": "cov_accum_kernel (v_mfma_f64), same batch", "achieved": tf,
                                            "frac": tf / FP64_MFMA_PEAK_TFLOPS, "avg_launch_ms": e0.elapsed_time(e1) / 2}
            s8 = torch.zeros_like(scratch)
            for _ in range(3):
                ops.cov_accum_i8(s8, h, report=False)
            out["roofline"]["error_bound"]["sigma_i8_vs_f64_entrywise_max"] = entrywise_err(s8, scratch)   # one full batch x 3, both routes
            del scratch, s8
        # ONE FULL batch through the engine's default route, for the sigma check against the oracle
        n_sample = h.shape[0]
        sample_dev = {k: batches[0][k][:n_sample] for k in ("h", "x", "q", "k")}
        sig_sample = engine.new_covs(shape, dev)
        engine.accumulate(sig_sample, sample_dev, shape)
        sample = {k: v.cpu() for k, v in sample_dev.items()}
        cases = [(f"{li} shared", layers[li], covs, gpu_out)]
        if kept is not None:
            kli, kcovs, kout = kept
            cases.append((f"{kli} distinct", layers[kli], kcovs, kout))
        else:       # (no distinct set: the previous timed layer's weights on the shared statistics)
            pli = ids[-2] if len(ids) > 1 else li
            pt, pm = headline_out[pli]
            pout = dict(pt)
            pout["mask"], pout["mlp_idx"] = pm, gpu_out["mlp_idx"]
            cases.append((f"{pli} shared", layers[pli], covs, pout))
        out["cpu_baseline"] = cpu_baseline(shape, cases, sample, n_texts * 2048, a.keep, ridges, sig_sample)
        out["full_size_parity_vs_oracle"] = out["cpu_baseline"].pop("full_size_parity_vs_oracle")
        del sig_sample, cases
        if shape["arch"] != "opt":
            out["next_rows"] = {"rope_gather": rope_gather_roofline(shape, a.keep, dev)}
    elif rank == 0:
        out["cpu_baseline"] = None
    kept = None
    del last, done
    if rank == 0 and world == 1 and not a.no_extra_legs and shape["arch"] != "opt":
        if i8:
            # (1) the faithful route: the SAME step loop with every covariance on v_mfma_f64 (SURVEY section 7's parity path) --
            #     and its compressed tensors against the headline leg's, layer by layer, at the full token count
            leg_ids = ids[:max(1, min(len(ids), a.extra_leg_steps))]
            ids64 = ids[:max(1, min(len(ids), a.extra_leg_steps, 6))]
            ops.COV_MODE = "f64"
            t64 = LaunchTimer()
            sec, outs64 = timed_steps(shape, adapter, ids64, batches, a.keep, n_texts, t64, pipelined, a.chain_depth)
            adapter.report_selection_margins()
            nl, fl, msl = t64.summary()
            out["value_f64_route"] = {"value": len(ids64) / sec, "ms_per_step": sec / len(ids64) * 1e3, "steps": len(ids64),
                                      "cov_kernel_tflops": fl / (msl * 1e-3) / 1e12,
                                      "cov_kernel_frac_of_fp64_peak": fl / (msl * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                      "avg_launch_ms": msl / nl}
            out["i8_vs_f64_outputs"] = compare_outputs(headline_out, outs64, n_texts * 2048)
            out["i8_vs_f64_outputs"]["distinct_sigmas"] = 1        # (the timed layers share one set of activations)
            if distinct is not None:
                out["i8_vs_f64_outputs"] = merge_comparisons(out["i8_vs_f64_outputs"], distinct)
            del outs64
            ops.COV_MODE = "i8"
            # (1b) the same steps one after the other on one stream: the per-kernel figures without the two streams' interference
            if pipelined:
                tp = LaunchTimer()
                sec, _ = timed_steps(shape, adapter, leg_ids, batches, a.keep, n_texts, tp, False)
                tp.price_i8(planes_per_batch)
                nl_s, fl_s, ms_s = tp.summary()
                n_d, d_fl, d_ms = tp.decomposition_summary()
                out["value_sequential"] = {"value": len(leg_ids) / sec, "ms_per_step": sec / len(leg_ids) * 1e3, "steps": len(leg_ids),
                                           "sigma_mlp_launch_ms": ms_s / nl_s,
                                           "sigma_mlp_frac": fl_s / (ms_s * 1e-3) / 1e12 * executed_fraction / INT8_MFMA_PEAK_TOPS,
                                           "decomposition": {"avg_ms_per_layer": d_ms / n_d, "achieved": d_fl / (d_ms * 1e-3) / 1e12,
                                                             "frac": d_fl / (d_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}}
            # (2) SiLU-gated sigma_mlp activations, what a real Llama MLP feeds the hook (LlamaAdapter.py:127-136): six planes
            gated = []
            for b, bt in enumerate(batches):
                gen = torch.Generator(device=dev).manual_seed(4242 + b)
                g = torch.nn.functional.silu(torch.randn(bt["h"].shape, generator=gen, device=dev, dtype=torch.float32))
                g.mul_(torch.randn(bt["h"].shape, generator=gen, device=dev, dtype=torch.float32))
                gated.append({"h": g.to(torch.bfloat16), "x": bt["x"], "q": bt["q"], "k": bt["k"]})
                del g
            out["value_gated"] = extra_leg(shape, adapter, leg_ids, gated, a.keep, n_texts, pipelined, dev, tokens)
            # (2b) the same data on the TRUNCATED product (MDG_I8_NO_EXACT: six planes, 15.1 executed plane pairs, the bound of round 3) --
            #      what the exact route replaced -- and that product with the tolerance factor at 64 (five planes + two columns on the
            #      fp64 column kernel; the factor does nothing on the exact route, which drops no plane pair)
            exact_before = ops.I8_EXACT
            ops.I8_EXACT = False
            try:
                half = leg_ids[:max(1, len(leg_ids) // 2)]
                trunc = extra_leg(shape, adapter, half, gated, a.keep, n_texts, pipelined, dev, tokens)
                with ops.i8_tolerance_scope(64.0 * tol):
                    loose = extra_leg(shape, adapter, half, gated, a.keep, n_texts, pipelined, dev, tokens)
            finally:
                ops.I8_EXACT = exact_before
            keys = ("value", "ms_per_step", "steps", "planes", "avg_launch_ms", "whole_call_ms", "error_bound")
            out["value_gated"]["truncated_product"] = {k: trunc[k] for k in keys}
            out["value_gated"]["truncated_product"]["tolerance_x64"] = {k: loose[k] for k in keys}
            del gated
            # (3) massive activations: four BOS-like columns (bulk 12-15 binades under three spikes per batch) in the residual
            #     stream statistic AND in the MLP statistic -- they leave the int8 launch alone, through the fp64 column kernel
            massive = []
            for b, bt in enumerate(batches):
                gen = torch.Generator(device=dev).manual_seed(777 + b)
                nb = dict(bt)
                for key, cols in (("h", (5, 129, shape["d_ff"] // 2 + 77, shape["d_ff"] - 1)), ("x", (3, 1000, 2533, shape["d"] - 2))):
                    t = bt[key].clone()
                    for i, c in enumerate(cols):
                        top = t[:, c].float().abs().max()
                        t[:, c] = (t[:, c].float() * 2.0 ** -(12 + i)).to(torch.bfloat16)
                        rows = torch.randperm(t.shape[0], device=dev, generator=gen)[:3]
                        t[rows, c] = (top * (1.0 + torch.rand(3, device=dev, generator=gen))).to(torch.bfloat16)
                    nb[key] = t
                massive.append(nb)
            out["value_massive"] = extra_leg(shape, adapter, leg_ids[:max(1, min(len(leg_ids), 6))], massive, a.keep, n_texts, pipelined, dev, tokens)
            out["value_massive"]["vs_value"] = out["value_massive"]["value"] / out["value"]
    if rank == 0 and distinct is not None and "i8_vs_f64_outputs" not in out:
        out["i8_vs_f64_outputs"] = distinct
    if rank == 0:
        print(json.dumps(_compact(out)))
    sharding.finalize()


def _compact(o):
    """Six significant digits are all these measurements carry; the line stays short enough for the driver's record to keep
    every field."""
    if isinstance(o, float):
        return float(f"{o:.6g}")
    if isinstance(o, dict):
        return {k: _compact(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_compact(v) for v in o]
    return o


def compare_outputs(headline, outs64, n_tokens):
    """The headline leg's compressed tensors (default int8 route) against the --cov-mode f64 leg's, same layers, full token count:
    selections (gathered rows, rotary mask) must be bit-identical, solved tensors agree to the stated tolerance."""
    res = {"layers": 0, "tokens": n_tokens, "up_identical": True, "gate_identical": True, "q_proj_identical": True,
           "k_proj_identical": True, "mask_identical": True, "down_max_rel": 0.0, "v_proj_max_rel": 0.0, "o_proj_max_rel": 0.0,
           "down_bf16_mismatch_frac": 0.0}
    for li, t64, m64, _ in outs64:
        if li not in headline:
            continue
        t8, m8 = headline[li]
        res["layers"] += 1
        for k in ("up", "gate", "q_proj", "k_proj"):
            if k in t8 and t8[k] is not None:
                res[k + "_identical"] = res[k + "_identical"] and bool(torch.equal(t8[k], t64[k]))
        res["mask_identical"] = res["mask_identical"] and bool(torch.equal(m8, m64))
        for k in ("down", "v_proj", "o_proj"):
            a8, a64 = t8[k].double(), t64[k].double()
            res[k + "_max_rel"] = max(res[k + "_max_rel"], float(((a8 - a64).abs().max() / a64.abs().max()).item()))
        res["down_bf16_mismatch_frac"] = max(res["down_bf16_mismatch_frac"], float((t8["down"] != t64["down"]).double().mean().item()))
    return res


def summarise_selection(report):
    """adapter.report_selection_margins() -> the bench line's short form: over the layers, the smallest relative margin of the MLP
    selection threshold (s_(k+1) - s_(k)) / s_(k), the largest first-order bound on what the covariance route's error (eps) can do
    to a score there, how many layers are certified (bound below margin for every pair of a selected and an unselected score)."""
    if not report:
        return None
    vals = list(report.values())
    return {"layers": len(vals), "certified": sum(1 for v in vals if v["certified"]), "eps": max(v["eps"] for v in vals),
            "margin_min": min(v["margin"] for v in vals), "score_bound_max": max(v["score_bound"] for v in vals)}


def merge_comparisons(x, y):
    out = dict(x)
    for k, v in y.items():
        if k.endswith("_identical"):
            out[k] = bool(x.get(k, True) and v)
        elif k in ("layers", "distinct_sigmas"):
            out[k] = x.get(k, 0) + v
        elif k.endswith("_max_rel") or k.endswith("_frac"):
            out[k] = max(x.get(k, 0.0), v)
        else:
            out[k] = v
    return out


def distinct_sigma_parity(shape, adapter, layer_ids, a, dev, n_texts, tokens):
    """VERDICT r3 item 1a.  The timed loops feed every layer the same 32 resident batches (the metric wants the inputs in HBM before
    the clock starts), so their i8-vs-f64 comparison checks ONE sigma against many weight sets.  Here every layer gets activations
    of its own -- seed (1234, layer, batch) for the values and a per-layer seed for the column scales, one batch at a time, at the
    full token count -- and both routes accumulate the SAME batches (int8 digit planes / v_mfma_f64), finalise and compress; the
    compressed tensors are compared and both routes' selection certificates recorded.  Returns (comparison, (layer, sigma_i8,
    gpu outputs) of the first layer for the CPU oracle's full-size parity check)."""
    res = {"layers": 0, "tokens": n_texts * 2048, "distinct_sigmas": 0, "up_identical": True, "gate_identical": True,
           "q_proj_identical": True, "k_proj_identical": True, "mask_identical": True, "down_max_rel": 0.0, "v_proj_max_rel": 0.0,
           "o_proj_max_rel": 0.0, "down_bf16_mismatch_frac": 0.0, "sigma_mlp_entrywise_max": 0.0}
    kept, margins = None, {"i8": {}, "f64": {}}
    mode_before = ops.COV_MODE
    from modegpt_amd.compression.compress_mlp import _fl32
    for li in layer_ids:
        covs = {m: engine.new_covs(shape, dev) for m in ("i8", "f64")}
        for b in range(a.batches):
            bt = engine.make_activation_batch(shape, tokens, seed=(1234 * 1000 + li) * 1000 + b, device=dev, scale_seed=977 + 7919 * (li + 1))
            for m in ("i8", "f64"):
                engine.accumulate(covs[m], bt, shape, mode=m)
            del bt
        outs = {}
        for m in ("i8", "f64"):
            engine.finalize(covs[m], n_texts)
            ops.COV_MODE = m                         # (which error bound the certificate is taken against)
            tensors, mask = engine.compress_layer(adapter, li, covs[m], a.keep, check=True)
            outs[m] = (dict(tensors), mask)
            margins[m].update(adapter.report_selection_margins())
        ops.COV_MODE = mode_before
        res = merge_comparisons(res, compare_outputs({li: outs["i8"]}, [(li, outs["f64"][0], outs["f64"][1], None)], n_texts * 2048))
        res["distinct_sigmas"] += 1
        res["sigma_mlp_entrywise_max"] = max(res["sigma_mlp_entrywise_max"], entrywise_err(covs["i8"]["mlp"], covs["f64"]["mlp"]))
        if kept is None:
            sc = ops.ridge_scores(covs["i8"]["mlp"], _fl32(engine.RECIPE_RIDGES["nystrom_ridge"]))
            gpu_out = dict(outs["i8"][0])
            gpu_out["mask"], gpu_out["mlp_idx"] = outs["i8"][1], ops.select_smallest_sorted(sc, int(shape["d_ff"] * a.keep))
            kept = (li, covs["i8"], gpu_out)
        del covs, outs
    res["selection_certificate"] = {m: summarise_selection(v) for m, v in margins.items()}
    return res, kept


def extra_leg(shape, adapter, ids, data, keep, n_texts, pipelined, dev, tokens):
    """The same step loop on other sigma_mlp / sigma_x data (value_gated, value_massive): layers/s, the sigma_mlp product launch
    timed alone and priced on the MFMAs it issued, the routes the device took and the bound it computed."""
    f = shape["d_ff"]
    before = ops.i8_route_counts(dev)
    step(shape, adapter, ids[0], data, keep, n_texts)                                   # warm-up (the six-plane kernel's first launch)
    adapter.report_selection_margins()                                                  # (not the timed layers')
    tg = LaunchTimer()
    sec, _ = timed_steps(shape, adapter, ids, data, keep, n_texts, tg, pipelined)
    sel = summarise_selection(adapter.report_selection_margins())
    after = ops.i8_route_counts(dev)
    st6, info = {}, {}
    scratch = torch.zeros(f, f, dtype=torch.float64, device=dev)
    used = ops.cov_accum_i8(scratch, data[0]["h"], mfma_stats=st6, route_info=info)
    # the whole sigma_mlp call (column maxima, split, route, product, and on the exact route the remainder kernel), timed alone
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record()
    for _ in range(3):
        ops.cov_accum_i8(scratch, data[0]["h"], report=False)
    w1.record()
    torch.cuda.synchronize()
    whole_call_ms = w0.elapsed_time(w1) / 3
    del scratch
    nl, _, msl = tg.summary()
    frac = st6["executed"] / st6["dense"] if st6.get("dense") else 1.0
    used_class = used
    if used and info.get("exact"):
        used = 3                                             # (the exact route: the nine-pair launch of the three top planes + the remainder kernel)
    pairs = {3: 9, 5: 15, 6: 21}.get(used, 15)
    tops_dense = pairs * nl * tokens * f * (f + 1) / (msl * 1e-3) / 1e12
    return {"value": len(ids) / sec, "ms_per_step": sec / len(ids) * 1e3, "steps": len(ids), "planes": used_class, "exact_route": bool(info.get("exact")),
            "avg_launch_ms": msl / nl, "whole_call_ms": whole_call_ms,
            "executed_fraction": frac, "achieved": tops_dense * frac, "frac": tops_dense * frac / INT8_MFMA_PEAK_TOPS,
            "routes": {k: after[k] - before[k] for k in after}, "error_bound": info.get("bound"), "fp64_columns_mlp": info.get("columns"),
            "selection_certificate": sel}


if __name__ == "__main__":
    main()
