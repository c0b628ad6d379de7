"""Several layers' decomposition chains in flight at once.

The reference's per-stage drivers walk the layers one after the other (compress_mlp.py:67-117, compress_vo.py:33-109).  A
layer's chain is a string of short dependent kernels between large GEMMs -- 191 single-workgroup Cholesky steps for the MLP
stage -- and leaves most of the chip idle in between; the next layer's chain does not depend on it.  `over_layers` keeps
`CHAIN_WIDTH` chains in flight, each on a stream of its own: the host enqueues layer i + 1 while layer i runs, reads layer
i - 1's status and writes its artefact meanwhile.  Same kernels, same inputs, same results, same order of the artefacts and
log lines; measured at Llama-3-8B shapes: 121 -> 97 ms per layer for two chains (scripts/probes/decomp_phases.py pair)."""
from __future__ import annotations

import os
from collections import deque
from typing import Callable, Sequence

import torch

from .. import ops
from ..model_utils import local_device

CHAIN_WIDTH = int(os.environ.get("MODEGPT_CHAIN_WIDTH", "2"))
_STREAMS = {}


def _streams(dev: torch.device, n: int):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    have = _STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


def _check(adapter, status) -> None:
    (getattr(adapter, "chain_status", None) or (lambda st: st.check()))(status)     # (a duck-typed adapter: read it now)


def over_layers(adapter, layers: Sequence[int], enqueue: Callable, retire: Callable) -> None:
    """enqueue(layer) -> result: the layer's kernels, nothing that waits for the device (runs inside an ops.DeferredStatus on
    the layer's stream); retire(layer, result): everything that does (save_layer, logging), called in layer order after the
    chain's status has been handed to adapter.chain_status.  An exception of enqueue / retire / the status check leaves with the
    chains already in flight joined into the caller's stream."""
    dev = torch.device(local_device())
    width = min(CHAIN_WIDTH, len(layers))
    if dev.type != "cuda" or width <= 1:
        for layer in layers:
            with ops.DeferredStatus(dev) as status:
                result = enqueue(layer)
            _check(adapter, status)
            retire(layer, result)
        return
    main = torch.cuda.current_stream(dev)
    streams = _streams(dev, width)
    inflight = deque()

    def retire_oldest():
        layer, result, status, st = inflight.popleft()
        main.wait_stream(st)            # the caller's stream (and whoever waits on it: the artefact copy) sees the results
        _check(adapter, status)
        retire(layer, result)

    try:
        for i, layer in enumerate(layers):
            if len(inflight) == width:
                retire_oldest()         # (its stream is the one this layer takes)
            st = streams[i % width]
            st.wait_stream(main)        # inputs produced on the caller's stream (the statistics, the weights)
            with torch.cuda.stream(st):
                with ops.DeferredStatus(dev) as status:
                    result = enqueue(layer)
            inflight.append((layer, result, status, st))
        while inflight:
            retire_oldest()
    finally:
        for _, _, _, st in inflight:
            main.wait_stream(st)
