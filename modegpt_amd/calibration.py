"""Calibration: run the model over the calibration batches while hooks accumulate the fp64 second-moment
matrices, plus Block-Influence scores (reference: src/calibration.py)."""
from __future__ import annotations

import logging
import random
from typing import List

import numpy as np
import torch

from . import ops
from .adapters.model_adapter import ModelAdapter
from .model_utils import calib_device, dtype_p

logger = logging.getLogger("MoDeGPT")

np.random.seed(1234)
random.seed(1234)

SEQ_LEN_NORMALISER = 2048  # calibration.py:141: the normaliser is n_texts * 2048 whatever the real length


def load_calibs(adapter: ModelAdapter, n_samples: int, batch_size: int, dataset: str = "wikitext",
                load_calibs_from="", calibs_save_path="", target_layers: List[int] = []):
    """calibration.py:18-36 -> (cov_mlp, cov_q, cov_k, cov_x, bi_scores); the four lists have one entry per
    layer, None for layers outside target_layers."""
    return _calibrate_model(adapter, n_samples=n_samples, batch_size=batch_size, dataset=dataset,
                            target_layers=target_layers)


@torch.no_grad()
def _calibrate_model(adapter: ModelAdapter, n_samples: int, batch_size: int, target_layers: List[int] = [],
                     dataset="wikitext"):
    model = adapter.model
    n_layers, n_heads, head_dim = adapter.n_layers, adapter.n_heads, adapter.head_dim
    blocks = adapter.get_transformer_blocks()
    if not target_layers:
        target_layers = list(range(n_layers))
    model.config.output_hidden_states = True
    if adapter.calibs is None:
        from .eval import load_calibration_texts
        adapter.calibs = load_calibration_texts(calib_size=n_samples, model=adapter.model, tokenizer=adapter.tokenizer,
                                                batch_size=batch_size, dataset=dataset)
    logger.info(f"Detected architecture: {adapter.arch}")
    logger.info(f"target_layers = {target_layers}")
    logger.info("Calibrating model")

    cov_mlp = [None] * n_layers
    cov_q = [None] * n_layers
    cov_k = [None] * n_layers
    cov_x = [None] * n_layers
    d_int = adapter.get_n_inner()
    for i in target_layers:  # calibration.py:82-96
        cov_mlp[i] = torch.zeros(d_int, d_int, dtype=dtype_p, device=calib_device)
        cov_q[i] = torch.zeros(n_heads, head_dim, head_dim, dtype=dtype_p, device=calib_device)
        cov_k[i] = torch.zeros(adapter.n_kv_heads, head_dim, head_dim, dtype=dtype_p, device=calib_device)
        cov_x[i] = torch.zeros(adapter.d_model, adapter.d_model, dtype=dtype_p, device=calib_device)

    handles = []
    for i in target_layers:
        adapter.register_hooks(i, blocks[i], cov_mlp_list=cov_mlp, cov_q_list=cov_q, cov_k_list=cov_k,
                               cov_x_list=cov_x, handles=handles, logger=logger)

    model.eval()
    bi_dev = torch.zeros(n_layers, dtype=torch.float64, device=calib_device)  # running sums stay on the GPU
    bi_scores = [0.0] * n_layers
    n_texts = 0
    for batch in adapter.calibs:
        n_texts += len(batch)
        out = model(batch, output_hidden_states=True)
        hs = out.hidden_states
        T = hs[0].shape[1]
        step = torch.zeros(n_layers, dtype=torch.float64, device=calib_device)
        for l in range(n_layers):  # calibration.py:118-124: sum_B (1 - cos) then mean over T
            ops.bi_accum(step[l:l + 1], hs[l], hs[l + 1])
        bi_dev += step / T
        del hs, out
    for h in handles:
        h.remove()
    bi_host = bi_dev.cpu().tolist()  # one sync for all layers (the reference syncs per layer per batch)
    for l in range(n_layers):
        bi_scores[l] = bi_host[l] / n_texts
    adapter.bi_scores = bi_scores

    scale = 1.0 / (n_texts * SEQ_LEN_NORMALISER)
    for i in target_layers:  # calibration.py:141-146, fused with the lower->upper mirror
        for buf in (cov_mlp[i], cov_x[i], cov_k[i], cov_q[i]):
            ops.cov_finalize(buf, scale)
    logger.info("Finished calibration and computed BI scores.")
    return cov_mlp, cov_q, cov_k, cov_x, bi_scores


# the reference names this function with two leading underscores
__calibrate_model = _calibrate_model
