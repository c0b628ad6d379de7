"""Model-free harness around the hot path: synthetic weights / activations of a named architecture, an in-memory
ModelAdapter, and one-layer drivers.  bench.py, __graft_entry__.smoke() and the end-to-end parity tests use it so
that what they time and check is the same `load_calibs`-hook kernels and the same `compress_nystrom` /
`compress_qk` / `compress_vo` functions a real run calls -- minus HF model forward, tokenisation and disk IO, which
the metric excludes."""
from __future__ import annotations

import math
import os
import types
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .adapters.CompressionConfig import CompressionConfig
from .adapters.model_adapter import (AttentionComponents, MLPComponents, MLPTensors, ModelAdapter, QKComponents,
                                     QKTensors, VOComponents, VOTensors)
from .compression.compress_mlp import compress_nystrom
from .compression.compress_qk import compress_qk
from .compression.compress_vo import compress_vo

# public model-card shapes
SHAPES = {
    "llama-3-8b": dict(arch="llama", n_layers=32, d=4096, d_ff=14336, n_heads=32, n_kv_heads=8, head_dim=128),
    "llama-2-7b": dict(arch="llama", n_layers=32, d=4096, d_ff=11008, n_heads=32, n_kv_heads=32, head_dim=128),
    "qwen3-14b": dict(arch="qwen3", n_layers=40, d=5120, d_ff=17408, n_heads=40, n_kv_heads=8, head_dim=128),
    "opt-125m": dict(arch="opt", n_layers=12, d=768, d_ff=3072, n_heads=12, n_kv_heads=12, head_dim=64),
    "tiny": dict(arch="llama", n_layers=2, d=256, d_ff=640, n_heads=4, n_kv_heads=2, head_dim=64),
}
# the tests.sh recipe of the reference (tests.sh:100-104)
RECIPE_RIDGES = dict(nystrom_ridge=1e-4, ridge_qk=1e-2, ridge_vo=1e-5)


def make_layer_weights(shape: dict, seed: int, device) -> Dict[str, torch.Tensor]:
    """bf16 N(0, 0.02^2) weights of one layer, seed 1234 + layer as in SURVEY.md 8d."""
    g = torch.Generator(device=device).manual_seed(seed)
    d, f, nh, nkv, hd = shape["d"], shape["d_ff"], shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]

    def w(r, c):
        return (torch.randn(r, c, device=device, generator=g) * 0.02).to(torch.bfloat16)

    out = {"up": w(f, d), "down": w(d, f), "q": w(nh * hd, d), "k": w(nkv * hd, d), "v": w(nkv * hd, d),
           "o": w(d, nh * hd)}
    if shape["arch"] != "opt":
        out["gate"] = w(f, d)
    return out


def make_activation_batch(shape: dict, tokens: int, seed: int, device, scale_seed: int = 977) -> Dict[str, torch.Tensor]:
    """One calibration batch's hook inputs: z * c_j, z ~ N(0,1), per-feature scale log-uniform[0.05, 2]
    (non-flat spectrum so selections are not tie-dominated), bf16.  scale_seed: the seed of the per-feature scales -- a property
    of the LAYER (the same for all its batches); give every layer its own to get layers with different statistics."""
    g = torch.Generator(device=device).manual_seed(seed)
    d, f, nh, nkv, hd = shape["d"], shape["d_ff"], shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]

    def a(feat, salt):
        gs = torch.Generator(device=device).manual_seed(scale_seed + salt)  # the scale is a property of the feature, not the batch
        c = torch.exp(torch.empty(feat, device=device).uniform_(math.log(0.05), math.log(2.0), generator=gs))
        return (torch.randn(tokens, feat, device=device, generator=g) * c).to(torch.bfloat16)

    return {"h": a(f, 1), "x": a(d, 2), "q": a(nh * hd, 3), "k": a(nkv * hd, 4)}


def new_covs(shape: dict, device) -> Dict[str, torch.Tensor]:
    d, f, nh, nkv, hd = shape["d"], shape["d_ff"], shape["n_heads"], shape["n_kv_heads"], shape["head_dim"]
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=device)  # noqa: E731
    return {"mlp": z(f, f), "x": z(d, d), "q": z(nh, hd, hd), "k": z(nkv, hd, hd)}


def accumulate(covs: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], shape: dict, mode: Optional[str] = None) -> None:
    """What the four hooks of one layer do for one calibration batch: one fused launch (the Llama / Qwen3 adapters
    defer sigma_x / sigma_q / sigma_k to the layer's last hook for exactly this); OPT's ReLU statistic stays separate.
    mode: None = ops.COV_MODE; "i8" routes sigma_mlp and sigma_x through the int8 digit-plane kernel."""
    if shape["arch"] == "opt":
        ops.cov_accum(covs["mlp"], batch["h"], relu=True)
        ops.cov_accum_multi([(covs["x"], batch["x"], 1), (covs["q"], batch["q"], shape["n_heads"]),
                             (covs["k"], batch["k"], shape["n_kv_heads"])], mode=mode)
        return
    ops.cov_accum_multi([(covs["mlp"], batch["h"], 1), (covs["x"], batch["x"], 1),
                         (covs["q"], batch["q"], shape["n_heads"]), (covs["k"], batch["k"], shape["n_kv_heads"])], mode=mode)


def finalize(covs: Dict[str, torch.Tensor], n_texts: int) -> None:
    for c in covs.values():
        ops.cov_finalize(c, 1.0 / (n_texts * 2048))


class _W:
    """Stands where the compressors expect an nn.Linear: only `.weight` / `.bias` are read."""
    def __init__(self, weight, bias=None):
        self.weight, self.bias = weight, bias


class TensorAdapter(ModelAdapter):
    """ModelAdapter over bare weight tensors.  `save_layer` keeps the artefacts in HBM (self.store) instead of
    torch.save-ing them, since the measured path excludes disk IO."""

    def __init__(self, shape: dict, layers: Dict[int, Dict[str, torch.Tensor]], config: Optional[CompressionConfig] = None):
        self.shape = shape
        self.layers = layers
        self.model = None
        self.tokenizer = None
        self.calibs = None
        self.model_config = types.SimpleNamespace(
            model_type=shape["arch"], num_hidden_layers=shape["n_layers"], num_attention_heads=shape["n_heads"],
            num_key_value_heads=shape["n_kv_heads"], hidden_size=shape["d"], intermediate_size=shape["d_ff"],
            head_dim=shape["head_dim"])
        self.config = config or CompressionConfig(**RECIPE_RIDGES)
        self.metrics = {}
        self.store: Dict[tuple, Dict[str, torch.Tensor]] = {}
        self.pending_status = []

    @property
    def arch(self) -> str:
        return self.shape["arch"]

    def save_layer(self, output_dir, suffix, weights, layer_idx):
        self.store[(layer_idx, suffix)] = weights

    def chain_status(self, status):
        """Results stay on the device, so nothing here has to wait for the stream: the statuses of the layers' chains are
        collected and read together by check_chains() (bench.py: after the last layer is enqueued)."""
        self.pending_status.append(status)

    def check_chains(self):
        pending, self.pending_status = self.pending_status, []
        for st in pending:
            st.check()

    def get_transformer_blocks(self):
        raise NotImplementedError("TensorAdapter has no nn.Module blocks")

    def register_hooks(self, *a, **k):
        raise NotImplementedError("TensorAdapter is fed through engine.accumulate")

    def compute_layer_energy(self, layer_idx, Ca=None):
        raise NotImplementedError

    def calibrate_model(self, n_samples, batch_size, target_layers, dataset="wikitext"):
        raise NotImplementedError

    def _w(self, i, k):
        t = self.layers[i].get(k)
        return None if t is None else _W(t)

    def get_mlp_components(self, layer_idx, expert_idx=None):
        return MLPComponents(block=None, up_proj=self._w(layer_idx, "up"), down_proj=self._w(layer_idx, "down"),
                             gate_proj=self._w(layer_idx, "gate"))

    def get_mlp_tensors(self, layer_idx, expert_idx=None):
        L = self.layers[layer_idx]
        return MLPTensors(up_proj=L["up"], down_proj=L["down"], gate_proj=L.get("gate"))

    def get_vo_components(self, layer_idx, expert_idx=None):
        return VOComponents(block=None, v_proj=self._w(layer_idx, "v"), o_proj=self._w(layer_idx, "o"))

    def get_vo_tensors(self, layer_idx, expert_idx=None):
        L = self.layers[layer_idx]
        return VOTensors(v_proj=L["v"], o_proj=L["o"])

    def get_qk_components(self, layer_idx, expert_idx=None):
        return QKComponents(block=None, query_proj=self._w(layer_idx, "q"), key_proj=self._w(layer_idx, "k"))

    def get_qk_tensors(self, layer_idx, expert_idx=None):
        L = self.layers[layer_idx]
        return QKTensors(query_proj=L["q"], key_proj=L["k"])

    def get_attn_components(self, layer_idx):
        return AttentionComponents(block=None, q_proj=self._w(layer_idx, "q"), k_proj=self._w(layer_idx, "k"),
                                   v_proj=self._w(layer_idx, "v"), o_proj=self._w(layer_idx, "o"))

    def get_qk_weights(self, layer_idx):
        L = self.layers[layer_idx]
        return L["q"], L["k"]

    def get_vo_weights(self, layer_idx):
        L = self.layers[layer_idx]
        return L["v"], L["o"]

    def replace_mlp_layers(self, layer_idx, new_up, new_down, new_gate=None, expert_idx=None):
        raise NotImplementedError

    def replace_attn_layers(self, layer_idx, new_q, new_k, new_v, new_o):
        raise NotImplementedError


OVERLAP_VO = os.environ.get("MODEGPT_OVERLAP_VO", "1") != "0"
_VO_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def compress_layer(adapter: TensorAdapter, layer_idx: int, covs: Dict[str, torch.Tensor], keep_ratio: float, check: bool = True):
    """mlp -> qk -> vo for one layer through the drop-in functions (fixed order of run_modegpt.py:128-151).
    Returns the layer's compressed tensors and rotary mask.  The chain enqueues without a host round trip; check=True reads
    its status (not positive definite / eigensolver not converged -> exception) before returning, check=False leaves that to
    a later adapter.check_chains() (bench.py: once, after the last layer)."""
    n = max(adapter.shape["n_layers"], layer_idx + 1)
    lst = lambda t: [t if i == layer_idx else None for i in range(n)]  # noqa: E731
    keep = [keep_ratio] * n
    # VO (one GEMM, a batched 128 x 128 eigensolve on 8 workgroups, skinny products: 8 ms of mostly latency) needs sigma_x only and
    # runs on a stream of its own beside the MLP chain, whose Cholesky steps leave most of the chip idle between their GEMMs
    dev = covs["x"].device
    main = torch.cuda.current_stream(dev)
    side = None
    if OVERLAP_VO:
        key = (dev.index, main.cuda_stream)          # (one per caller's stream: two layers' chains stay independent)
        side = _VO_STREAMS.get(key)
        if side is None:
            side = _VO_STREAMS[key] = torch.cuda.Stream(device=dev)
    if side is not None:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            compress_vo(adapter=adapter, cov=lst(covs["x"]), keep_ratios=keep, target_layers=[layer_idx])
        covs["x"].record_stream(side)
    compress_nystrom(adapter=adapter, cov=lst(covs["mlp"]), keep_ratios=keep, target_layers=[layer_idx])
    masks = compress_qk(adapter=adapter, cov=(lst(covs["q"]), lst(covs["k"])), keep_ratios=keep,
                        target_layers=[layer_idx])
    if side is None:
        compress_vo(adapter=adapter, cov=lst(covs["x"]), keep_ratios=keep, target_layers=[layer_idx])
    else:
        main.wait_stream(side)
        for t in adapter.store[(layer_idx, "vo")].values():
            t.record_stream(main)        # allocated on the side stream, used on the caller's from here on
    if check:
        adapter.check_chains()
    out = {}
    for suffix in ("mlp", "qk", "vo"):
        out.update(adapter.store[(layer_idx, suffix)])
    return out, (masks[0] if masks else None)
