"""Calibration pass: one sweep of the model over the calibration batches during which adapter hooks stream every
target layer's activations into the HIP covariance kernel, plus Block-Influence scores from the hidden states.

Public entry point and return contract follow the reference (src/calibration.py:18-36):
    load_calibs(adapter, n_samples, batch_size, dataset, load_calibs_from, calibs_save_path, target_layers)
        -> (cov_mlp, cov_q, cov_k, cov_x, bi_scores)
each cov_* a list with one fp64 device tensor per layer (None outside target_layers), bi_scores a list of floats
for ALL layers.  Internals differ: sigma buffers are lower-triangular until one fused mirror+normalise pass at the
end, and the BI running sums stay on the device (one host sync per calibration, not one per layer per batch).
"""
from __future__ import annotations

import logging
import random
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .adapters.model_adapter import ModelAdapter
from .model_utils import dtype_p, local_device

logger = logging.getLogger("MoDeGPT")

np.random.seed(1234)
random.seed(1234)

# The normaliser is n_texts * 2048 whatever the real sequence length is (src/calibration.py:141).
TOKENS_PER_TEXT_NORMALISER = 2048


class SigmaBuffers:
    """The four statistic families of the target layers (shapes: src/calibration.py:82-96)."""

    KINDS = ("mlp", "q", "k", "x")

    def __init__(self, adapter: ModelAdapter, target_layers: Sequence[int], device=None):
        L = adapter.n_layers
        hd, d, f = adapter.head_dim, adapter.d_model, adapter.get_n_inner()
        shapes = {"mlp": (f, f), "q": (adapter.n_heads, hd, hd), "k": (adapter.n_kv_heads, hd, hd), "x": (d, d)}
        self.lists: Dict[str, List[Optional[torch.Tensor]]] = {k: [None] * L for k in self.KINDS}
        self.layers = list(target_layers)
        device = device or local_device()          # this rank's own GPU (calib_device upstream: src/calibration.py:83)
        for i in self.layers:
            for kind, shp in shapes.items():
                self.lists[kind][i] = torch.zeros(*shp, dtype=dtype_p, device=device)

    def finalize(self, n_texts: int) -> None:
        """sigma <- sigma / (n_texts * 2048), lower triangle mirrored into the upper (src/calibration.py:141-146)."""
        scale = 1.0 / (n_texts * TOKENS_PER_TEXT_NORMALISER)
        for i in self.layers:
            for kind in self.KINDS:
                ops.cov_finalize(self.lists[kind][i], scale)


class BlockInfluence:
    """bi[l] = (1/n_texts) * sum over batches of mean_T sum_B (1 - cos(h_l, h_{l+1}))  (src/calibration.py:118-136)."""

    def __init__(self, n_layers: int, device=None):
        self.n_layers = n_layers
        device = device or local_device()
        self.acc = torch.zeros(n_layers, dtype=torch.float64, device=device)

    def add_batch(self, hidden_states) -> None:
        T = hidden_states[0].shape[1]
        step = torch.zeros_like(self.acc)
        for l in range(self.n_layers):
            ops.bi_accum(step[l:l + 1], hidden_states[l], hidden_states[l + 1])
        self.acc += step / T

    def scores(self, n_texts: int) -> List[float]:
        return [v / n_texts for v in self.acc.cpu().tolist()]


class StopForward(Exception):
    """Raised by a forward hook behind the last layer a rank owns: the rest of the model forward is not needed there."""


def _stop_forward(module, args, output):
    raise StopForward()


def load_calibs(adapter: ModelAdapter, n_samples: int, batch_size: int, dataset: str = "wikitext",
                load_calibs_from="", calibs_save_path="", target_layers: List[int] = []):
    return _calibrate_model(adapter, n_samples=n_samples, batch_size=batch_size, dataset=dataset,
                            target_layers=target_layers)


@torch.no_grad()
def _calibrate_model(adapter: ModelAdapter, n_samples: int, batch_size: int, target_layers: List[int] = [],
                     dataset="wikitext"):
    model = adapter.model
    targets = list(target_layers) if target_layers else list(range(adapter.n_layers))
    if getattr(adapter, "calib_no_hooks", False):      # a sharded rank that owns no layer of this chunk (BI only, or nothing)
        targets = []
    if adapter.calibs is None:
        from .eval import load_calibration_texts
        adapter.calibs = load_calibration_texts(calib_size=n_samples, model=model, tokenizer=adapter.tokenizer,
                                                batch_size=batch_size, dataset=dataset)
    logger.info(f"Detected architecture: {adapter.arch}")
    logger.info(f"target_layers = {targets}")
    logger.info("Calibrating model")

    sig = SigmaBuffers(adapter, targets)
    bi = BlockInfluence(adapter.n_layers)
    blocks = adapter.get_transformer_blocks()
    handles: list = []
    for i in targets:
        adapter.register_hooks(i, blocks[i], cov_mlp_list=sig.lists["mlp"], cov_q_list=sig.lists["q"],
                               cov_k_list=sig.lists["k"], cov_x_list=sig.lists["x"], handles=handles, logger=logger)
    # A sharded run (run_modegpt.compress_chunk) sets two attributes on the adapter: calib_stop_after = the last layer this rank
    # needs activations of -- the forward is cut right behind it -- and calib_want_bi = whether this rank is the one that
    # computes the BI scores (they need every layer's hidden state, i.e. the full forward; the others receive them).
    stop_after = getattr(adapter, "calib_stop_after", None)
    want_bi = getattr(adapter, "calib_want_bi", True)
    if want_bi:
        stop_after = None
    if stop_after is not None:
        handles.append(blocks[stop_after].register_forward_hook(_stop_forward))
    hidden_states_before = getattr(model.config, "output_hidden_states", False)
    model.config.output_hidden_states = bool(want_bi)
    model.eval()
    n_texts = n_tokens = 0
    try:
        for batch in adapter.calibs:
            n_texts += len(batch)
            n_tokens += int(batch.numel()) if hasattr(batch, "numel") else 0
            try:
                out = model(batch, output_hidden_states=bool(want_bi))
            except StopForward:
                continue
            if want_bi:
                bi.add_batch(out.hidden_states)
            del out
    finally:
        for h in handles:
            h.remove()
        model.config.output_hidden_states = hidden_states_before     # (the caller's model goes back as it came)
    bi_scores = bi.scores(n_texts) if want_bi else None
    adapter.bi_scores = bi_scores
    adapter.calib_tokens = n_tokens       # (how many tokens each statistic summed over: the fp64 route's rounding bound scales with it)
    sig.finalize(n_texts)
    if ops.COV_MODE == "i8":   # the route of every large-statistic launch was picked on the device; read the tally once
        adapter.cov_routes = ops.i8_route_counts(reset=True)
        logger.info(f"covariance routes (int8 five planes / six planes / fp64 fallback): {adapter.cov_routes}")
    logger.info("Finished calibration and computed BI scores.")
    return sig.lists["mlp"], sig.lists["q"], sig.lists["k"], sig.lists["x"], bi_scores


__calibrate_model = _calibrate_model  # the reference's (name-mangled) spelling
