"""Command-line driver.

    python -m modegpt_amd.run_modegpt <CompressionConfig flags>      (or: python -m src.run_modegpt ...)
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m modegpt_amd.run_modegpt ...   (layer-sharded)

The sequence is the reference's (src/run_modegpt.py:71-196): baseline perplexity -> for each chunk of 48 layers
{load_calibs -> allocate_global_sparsity -> compress_nystrom -> compress_qk -> compress_vo} -> convert_model ->
patch_config -> save -> reload -> compressed perplexity -> metrics.  Under torch.distributed every rank compresses
its contiguous share of each chunk and one all-gather reassembles the artefacts (sharding.py).
"""
from __future__ import annotations

import gc
import logging
import os
from typing import List, Optional

import torch

from . import sharding
from .adapters.CompressionConfig import CompressionConfig
from .adapters.model_adapter import ModelAdapter
from .calibration import load_calibs
from .compression.compress_mlp import compress_nystrom
from .compression.compress_qk import compress_qk
from .compression.compress_vo import compress_vo
from .compression_utils import allocate_global_sparsity
from .eval import compute_perplexity
from .model_utils import reload_compressed_model, save_compressed_model, start_memory_usage_worker

LAYERS_PER_STEP = 48          # src/run_modegpt.py:107
DEFAULT_ORDER = "mlp,qk,vo"   # upstream leaves --order unset and then fails on `"mlp" in None`

logger = logging.getLogger("MoDeGPT")


def _setup_logging() -> None:
    logger.setLevel(logging.INFO)
    if logger.handlers:
        return
    fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    os.makedirs("logs", exist_ok=True)
    for handler in (logging.StreamHandler(), logging.FileHandler("logs/run_modegpt.log")):
        handler.setFormatter(fmt)
        logger.addHandler(handler)


def _free() -> None:
    gc.collect()
    torch.cuda.empty_cache()


def _share_bi_scores(adapter, bi_scores, src: int, rank: int):
    """BI scores from the rank that ran the full forward to everybody (the model and the calibration samples do not change between
    chunks, so later chunks reuse them: upstream recomputes identical values, run_modegpt.py:112-118)."""
    import torch.distributed as dist
    cached = getattr(adapter, "bi_scores_cached", None)
    if cached is not None:
        return cached
    dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
    t = torch.zeros(adapter.n_layers, dtype=torch.float64, device=dev)
    if rank == src:
        t.copy_(torch.tensor(bi_scores, dtype=torch.float64))
    dist.broadcast(t, src=src)
    adapter.bi_scores_cached = adapter.bi_scores = t.cpu().tolist()
    return adapter.bi_scores_cached


def compress_chunk(adapter: ModelAdapter, config: CompressionConfig, chunk: List[int], rank: int, world: int):
    """Calibrate + compress the layers of `chunk` this rank owns; returns the chunk's rotary masks in layer order."""
    mine = sharding.my_layers(chunk, rank, world)
    if world > 1:
        # Every rank takes all samples through the model, but only as far as its own layers reach (the hooks of a layer need the
        # forward up to that layer, nothing behind it).  The BI scores need every hidden state: the LAST rank -- whose block ends
        # the chunk and which owns the fewest layers under the balanced partition -- runs the forward to the end and computes
        # them once; the others receive them below (a 8 n_layers-byte control message, not a data-path exchange).
        bi_rank = world - 1
        adapter.calib_want_bi = rank == bi_rank and getattr(adapter, "bi_scores_cached", None) is None
        adapter.calib_stop_after = mine[-1] if mine else chunk[0]
        adapter.calib_no_hooks = not mine          # (`target_layers=[]` means "all layers" upstream: say "none" explicitly)
    try:
        cov_mlp, cov_q, cov_k, cov_x, bi_scores = load_calibs(
            adapter=adapter, n_samples=config.calib_size, batch_size=config.calibs_batch_size, dataset=config.dataset,
            target_layers=mine)
    finally:
        # the three switches belong to THIS call: a later load_calibs on the same adapter (another trial, a one-rank re-run, a test
        # reusing the adapter) must get the full forward, its hooks and its BI scores again
        for name in ("calib_want_bi", "calib_stop_after", "calib_no_hooks"):
            if hasattr(adapter, name):
                delattr(adapter, name)
    if world > 1:
        bi_scores = _share_bi_scores(adapter, bi_scores, bi_rank, rank)
    keep = allocate_global_sparsity(bi_scores, compression_ratio=config.compression_ratio,
                                    smoothing=config.sparsity_smoothing, max_sparsity=config.max_sparsity,
                                    adapter=adapter)
    stages = config.order or DEFAULT_ORDER
    masks = []
    if "mlp" in stages:
        compress_nystrom(adapter=adapter, cov=cov_mlp, keep_ratios=keep, target_layers=mine)
    if "qk" in stages:
        masks = compress_qk(adapter=adapter, cov=(cov_q, cov_k), keep_ratios=keep, target_layers=mine)
    if "vo" in stages:
        compress_vo(adapter=adapter, cov=cov_x, keep_ratios=keep, target_layers=mine)
    del cov_mlp, cov_q, cov_k, cov_x
    _free()
    report = getattr(adapter, "report_selection_margins", None)     # (a duck-typed adapter has none)
    if report is not None:
        report(logger)                          # certificates of this chunk's MLP rank selections -> metrics["mlp_selection"], warnings
    return sharding.gather_layer_artifacts(adapter, chunk, mine, masks, rank, world)


@torch.no_grad()
def main(trial=None, config: Optional[CompressionConfig] = None):
    _setup_logging()
    _free()
    start_memory_usage_worker()
    config = config or CompressionConfig.from_args()
    print(config.to_dict())
    rank, world = sharding.init_from_env()
    if world > 1:   # blocks balanced against the forward a rank runs to reach them (measured share)
        os.environ.setdefault("MODEGPT_SHARD_FORWARD_SHARE", "0.32")

    model, tokenizer = reload_compressed_model(config.model)
    adapter = ModelAdapter.from_model(model=model, tokenizer=tokenizer)
    adapter.config = config
    adapter.async_artifacts(True)       # layer artefacts through a background writer; convert_model / the gather flush it
    adapter.hold_artifacts(world > 1)   # sharded: the all-gather's send buffer is packed from the tensors save_layer holds
    if rank == 0:   # (every rank holds the model; the metric is rank 0's business)
        adapter.metrics["baseline-ppl"] = compute_perplexity(model, tokenizer, dataset=config.dataset, adapter=adapter)
        logger.info(f"Baseline ppl: {adapter.metrics['baseline-ppl']}")
    _free()

    rotary_masks = []
    for first in range(0, adapter.n_layers, LAYERS_PER_STEP):
        chunk = list(range(first, min(adapter.n_layers, first + LAYERS_PER_STEP)))
        rotary_masks.extend(compress_chunk(adapter, config, chunk, rank, world))

    # All artefacts of all layers are now in temp_storage_dir as rank 0 needs them (its own + the gathered ones, the
    # latter written by rank 0 alone: sharding.gather_layer_artifacts).  One barrier closes the collective phase -- no rank
    # is still packing / writing when rank 0 starts reading -- and the process group is released here, so that the other
    # ranks do not sit in a collective (and its timeout) while rank 0 converts, saves and evaluates.
    adapter.flush_artifacts()
    sharding.finalize()
    if rank != 0:  # rank 0 alone writes the checkpoint
        return None
    save_dir = os.path.join(config.output_dir, "model")
    adapter.convert_model(saved_layers_dir=config.temp_storage_dir)
    adapter.patch_config()
    save_compressed_model(adapter, rotary_masks=rotary_masks, save_dir=save_dir, source_model_name=config.model)
    # the reference's flow: reload the checkpoint through the modeling file shipped with it (trust_remote_code)
    del model, tokenizer
    _free()
    adapter.model, adapter.tokenizer = reload_compressed_model(save_dir)
    ppl = compute_perplexity(adapter.model, adapter.tokenizer, dataset=config.dataset, adapter=adapter)
    adapter.metrics[f"ppl-{config.dataset}"] = ppl
    adapter.save_metrics()
    logger.info(f"Compressed (PPL): {ppl}")
    return ppl


if __name__ == "__main__":
    main()
