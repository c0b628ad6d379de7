"""Driver: `python -m modegpt_amd.run_modegpt` (or `python -m src.run_modegpt` through the src/ shim) with the
reference's CompressionConfig flags.  Same sequence as src/run_modegpt.py:71-196 -- baseline perplexity,
layer-chunk loop {load_calibs -> allocate_global_sparsity -> compress_nystrom -> compress_qk -> compress_vo},
convert_model, patch_config, save, reload, compressed perplexity -- plus layer sharding when launched under
torch.distributed (one process per GPU; see sharding.py)."""
from __future__ import annotations

import gc
import logging
import os

import torch

from .adapters.CompressionConfig import CompressionConfig
from .adapters.model_adapter import ModelAdapter
from .calibration import load_calibs
from .compression.compress_mlp import compress_nystrom
from .compression.compress_qk import compress_qk
from .compression.compress_vo import compress_vo
from .compression_utils import allocate_global_sparsity
from .eval import compute_perplexity
from .model_utils import reload_compressed_model, save_compressed_model, start_memory_usage_worker
from . import sharding

logger = logging.getLogger("MoDeGPT")
logger.setLevel(logging.INFO)
if not logger.handlers:
    _fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    _console = logging.StreamHandler()
    _console.setFormatter(_fmt)
    logger.addHandler(_console)
    os.makedirs("logs", exist_ok=True)
    _file = logging.FileHandler("logs/run_modegpt.log")
    _file.setFormatter(_fmt)
    logger.addHandler(_file)

LAYERS_PER_STEP = 48  # run_modegpt.py:107


@torch.no_grad()
def main(trial=None, config: CompressionConfig | None = None):
    gc.collect()
    torch.cuda.empty_cache()
    start_memory_usage_worker()
    if not config:
        config = CompressionConfig.from_args()
    print(config.to_dict())
    rank, world = sharding.init_from_env()

    model, tokenizer = reload_compressed_model(config.model)
    adapter = ModelAdapter.from_model(model=model, tokenizer=tokenizer)
    adapter.config = config

    baseline_ppl = compute_perplexity(model, tokenizer, dataset=config.dataset, adapter=adapter)
    logger.info(f"Baseline ppl: {baseline_ppl}")
    adapter.metrics["baseline-ppl"] = baseline_ppl
    torch.cuda.empty_cache()

    order = config.order or "mlp,qk,vo"  # upstream: `"mlp" in None` raises when --order is omitted
    n_layers = adapter.n_layers
    save_dir = os.path.join(config.output_dir, "model")
    rotary_masks = []
    for start in range(0, n_layers, LAYERS_PER_STEP):
        chunk = list(range(start, min(n_layers, start + LAYERS_PER_STEP)))
        mine = sharding.my_layers(chunk, rank, world)      # this rank's share of the chunk (all of it when world == 1)
        cov_mlp, cov_q, cov_k, cov_x, bi_scores = load_calibs(
            adapter=adapter, n_samples=config.calib_size, batch_size=config.calibs_batch_size, dataset=config.dataset,
            target_layers=mine)
        keep = allocate_global_sparsity(bi_scores, compression_ratio=config.compression_ratio,
                                        smoothing=config.sparsity_smoothing, max_sparsity=config.max_sparsity,
                                        adapter=adapter)
        rms = []
        if "mlp" in order:
            compress_nystrom(adapter=adapter, cov=cov_mlp, keep_ratios=keep, target_layers=mine)
        if "qk" in order:
            rms = compress_qk(adapter=adapter, cov=(cov_q, cov_k), keep_ratios=keep, target_layers=mine)
        if "vo" in order:
            compress_vo(adapter=adapter, cov=cov_x, keep_ratios=keep, target_layers=mine)
        # one all-gather reassembles the chunk: layer artefacts land in temp_storage_dir on every rank's view
        rotary_masks.extend(sharding.gather_layer_artifacts(adapter, chunk, mine, rms, rank, world))
        cov_mlp = cov_q = cov_k = cov_x = None
        gc.collect()
        torch.cuda.empty_cache()

    if rank != 0:
        sharding.finalize()
        return None
    adapter.convert_model(saved_layers_dir=config.temp_storage_dir)
    adapter.patch_config()
    save_compressed_model(adapter, rotary_masks=rotary_masks, save_dir=save_dir, source_model_name=config.model)
    del model, tokenizer
    torch.cuda.empty_cache()
    gc.collect()
    model, tokenizer = reload_compressed_model(save_dir)
    adapter.model, adapter.tokenizer = model, tokenizer
    compressed_ppl = compute_perplexity(model, tokenizer, dataset=config.dataset, adapter=adapter)
    adapter.metrics[f"ppl-{config.dataset}"] = compressed_ppl
    adapter.save_metrics()
    logger.info(f"Compressed (PPL): {compressed_ppl}")
    sharding.finalize()
    return compressed_ppl


if __name__ == "__main__":
    main()
