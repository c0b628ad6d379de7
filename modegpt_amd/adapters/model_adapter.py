"""The model-adapter plug-in surface (reference: src/adapters/model_adapter.py).

Same class / method / property names and argument meanings, so adapters written against the reference ABC
drop in.  What differs is underneath: the statistic hooks hand the activation's device pointer to the HIP
covariance kernel (ops.cov_accum -> mdg_cov_accum) instead of upcasting to fp64 and calling torch matmul.
Between hook calls a sigma buffer holds only its LOWER triangle; `load_calibs` mirrors and normalises it
once at the end (mdg_cov_finalize).
"""
from __future__ import annotations

import logging
import os
from abc import ABC, abstractmethod
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from . import metrics as _metrics
from .CompressionConfig import CompressionConfig
from .components import (AttentionComponents, MLPComponents, MLPTensors, QKComponents, QKTensors,  # noqa: F401
                         VOComponents, VOTensors)


def build_metrics(_all_metrics: dict) -> dict:
    """Kept for callers of the reference's helper; the store itself lives in adapters/metrics.py."""
    m = _metrics.new_run()
    _all_metrics[m["RunName"]] = m
    return m


class ModelAdapter(ABC):
    _metrics: dict = {}

    def __init__(self, model: nn.Module, tokenizer=None):
        self.model = model
        self.model_config = model.config
        self.config: CompressionConfig = CompressionConfig()
        self.tokenizer = tokenizer
        self.calibs = None
        ModelAdapter.load_metrics()
        self.metrics = _metrics.new_run()

    # ---- construction (model_adapter.py:118-135) ----
    @staticmethod
    def from_model(model: nn.Module, tokenizer) -> "ModelAdapter":
        from .LlamaAdapter import LlamaAdapter
        from .OPTAdapter import OPTAdapter
        from .QwenAdapter import QwenAdapter
        inner = getattr(model, "model", None)
        if inner is not None and hasattr(inner, "decoder"):
            return OPTAdapter(model, tokenizer=tokenizer)
        if inner is not None and hasattr(inner, "layers"):
            if "qwen3" in (getattr(model.config, "model_type", None) or ""):
                return QwenAdapter(model, tokenizer=tokenizer)
            return LlamaAdapter(model, tokenizer=tokenizer)
        raise RuntimeError("Unsupported model architecture")

    # ---- metrics: thin delegates to adapters/metrics.py ----
    @staticmethod
    def load_metrics(path="./metrics/metrics.json"):
        _metrics.load(path)
        ModelAdapter._metrics = _metrics.ALL_RUNS

    @staticmethod
    def save_metrics_static(path="./metrics/metrics.json", backup_dir="./metrics/backups/",
                            jsons_path="./metrics/jsons/", run_metrics: Optional[dict] = None):
        _metrics.save(path, backup_dir, jsons_path, run_metrics)

    def save_metrics(self, path="./metrics/metrics.json", backup_dir="./metrics/backups/"):
        _metrics.save(path=path, backup_dir=backup_dir, run_metrics=self.metrics)

    def chain_status(self, status) -> None:
        """Called by compress_nystrom / compress_vo with the ops.DeferredStatus of a layer's kernel chain, right before the
        layer's artefact is saved: the default reads it now -- the save waits for the stream anyway, so this is the chain's one
        host round trip (upstream raises from inside torch.linalg.cholesky at the same point, compress_mlp.py:20,56).  An adapter
        that keeps results on the device (engine.TensorAdapter) collects the statuses and checks them later."""
        status.check()

    # ---- the certificate of the MLP rank selections (not upstream; north_star: "rank selections bit-identical") ----
    def selection_margin(self, layer_idx: int, out8, eps: float) -> None:
        """compress_nystrom hands over ops.select_margin's 8 numbers (still on the device) for layer `layer_idx`."""
        self.__dict__.setdefault("_selection_margins", {})[int(layer_idx)] = (out8, float(eps))

    def report_selection_margins(self, log=None) -> dict:
        """Reads the recorded certificates (one 64-byte copy per layer; call it where the host waits for the chains anyway), writes
        them to metrics["mlp_selection"] = {layer: {"margin", "score_bound", "eps", "eps_certifiable", "scores_at_risk", "certified"}}
        and WARNS for every layer whose selection the covariance error bound cannot certify: there the k-th and (k+1)-th ridge
        scores sit closer than the bound on what the route's sigma error can do to them, and another accumulation order / route /
        the reference's CPU arithmetic may select a different column set (compress_mlp.py:45-47)."""
        log = log or logging.getLogger("MoDeGPT")
        pending = self.__dict__.pop("_selection_margins", {})
        report = {}
        for layer in sorted(pending):
            out8, eps = pending[layer]
            m = ops.decode_margin(out8.cpu().tolist(), eps)
            report[layer] = {k: m[k] for k in ("margin", "score_bound", "eps", "eps_certifiable", "scores_at_risk", "certified")}
            if not m["certified"]:
                log.warning(f"[MLP] Layer {layer}: rank selection NOT certified -- threshold margin {m['margin']:.3e} against a score "
                            f"perturbation bound {m['score_bound']:.3e} (covariance error bound eps = {eps:.2e}); "
                            f"{m['scores_at_risk']} scores within reach of the threshold")
        if report:
            if not isinstance(getattr(self, "metrics", None), dict):
                self.metrics = {}
            store = self.metrics.setdefault("mlp_selection", {})
            store.update({str(k): v for k, v in report.items()})
        return report

    # ---- reconstruction: per-(layer, stage) artefacts and the final swap (model_adapter.py:184-237) ----
    def save_layer(self, output_dir: str, suffix: str, weights: dict, layer_idx):
        """torch.save({name: bf16 tensor}) to <output_dir>/layer_<i>_<suffix>; env vars in the path expand.  The file is in
        place when this returns -- unless the run switched the background writer on (async_artifacts(True): run_modegpt does),
        in which case it is in place after flush_artifacts(), which every reader of the directory here calls first."""
        output_dir = os.path.expandvars(output_dir)
        os.makedirs(output_dir, exist_ok=True)
        path = os.path.join(output_dir, f"layer_{layer_idx}_{suffix}")
        held = getattr(self, "_held_artifacts", None)
        if held is not None:                     # a sharded run: the gather packs its send buffer from these, not from the files
            held.setdefault(int(layer_idx), {}).update(weights)
        writer = getattr(self, "_artifact_writer", None)
        if writer is not None:
            writer.submit(path, weights)
        else:
            torch.save(weights, path)

    def hold_artifacts(self, enable: bool = True) -> None:
        """Keep a reference to every tensor handed to save_layer (on its device) until take_held_artifacts(layer) collects it:
        sharding.gather_layer_artifacts packs the all-gather's send buffer from them instead of reading the files back.  ~0.3 GB
        per layer at Llama-3-8B / 30 %, for the layers of one chunk a rank owns."""
        self._held_artifacts = {} if enable else None

    def take_held_artifacts(self, layer_idx: int) -> dict:
        held = getattr(self, "_held_artifacts", None)
        return {} if held is None else held.pop(int(layer_idx), {})

    def async_artifacts(self, enable: bool = True) -> None:
        """Layer artefacts through a background writer (artifact_io.ArtifactWriter): save_layer enqueues the device-to-host copy
        and returns, the next layer's kernels run meanwhile.  Off by default: a caller that reads a file right after save_layer
        finds it."""
        self.flush_artifacts()
        if enable and os.environ.get("MODEGPT_ASYNC_SAVE", "1") != "0":
            from ..artifact_io import ArtifactWriter
            self._artifact_writer = ArtifactWriter()
        else:
            self._artifact_writer = None

    def flush_artifacts(self) -> None:
        writer = getattr(self, "_artifact_writer", None)
        if writer is not None:
            writer.flush()

    @torch.no_grad()
    def convert_model(self, saved_layers_dir: str = "./compressed_output/layers/", suffixes=("mlp", "qk", "vo"),
                      device: str = "cuda"):
        """Swap every layer's Linears for bias-free bf16 Linears built from the saved artefacts."""
        saved_layers_dir = os.path.expandvars(saved_layers_dir)
        self.flush_artifacts()

        def linear_of(w: Tensor) -> nn.Linear:
            lin = nn.Linear(w.shape[1], w.shape[0], bias=False, device=device, dtype=torch.bfloat16)
            lin.weight.data.copy_(w.to(torch.bfloat16))
            return lin

        for suffix in suffixes:
            for i in range(self.n_layers):
                art = torch.load(os.path.join(saved_layers_dir, f"layer_{i}_{suffix}"), map_location=device)
                if suffix == "mlp":
                    gate = art.get("gate")
                    self.replace_mlp_layers(i, new_up=linear_of(art["up"]), new_down=linear_of(art["down"]),
                                            new_gate=None if gate is None else linear_of(gate))
                elif suffix == "qk":
                    self.replace_attn_layers(i, new_q=linear_of(art["q_proj"]), new_k=linear_of(art["k_proj"]),
                                             new_v=None, new_o=None)
                elif suffix == "vo":
                    self.replace_attn_layers(i, new_q=None, new_k=None, new_v=linear_of(art["v_proj"]),
                                             new_o=linear_of(art["o_proj"]))

    # ---- in-place slicing (model_adapter.py:394-542; upstream's call sites are commented out in favour of the
    #      save_layer -> convert_model route, the methods stay part of the adapter surface) ----
    @staticmethod
    def _swap_in(weight: Tensor, bias_from=None, bias_values: Optional[Tensor] = None) -> nn.Linear:
        """bf16 Linear on the GPU holding `weight` [out, in]; carries a bias only when `bias_from` (the module it
        replaces) has one, initialised from `bias_values` when given."""
        has_bias = bias_from is not None and getattr(bias_from, "bias", None) is not None
        lin = nn.Linear(weight.shape[1], weight.shape[0], bias=has_bias, device="cuda", dtype=torch.bfloat16)
        lin.weight.data.copy_(weight.to(torch.bfloat16))
        if has_bias and bias_values is not None:
            lin.bias.data.copy_(bias_values)
        return lin

    @torch.no_grad()
    def slice_gate_dims(self, layer_idx: int, up_weights: Tensor, down_weights: Tensor, gate_weights: Optional[Tensor],
                        new_bias_u: Optional[Tensor], new_bias_g: Optional[Tensor], bias: bool = True,
                        expert_idx: Optional[int] = None):
        """Put compressed MLP weights ([out, in] each) straight into the block.  up/gate biases come from the caller
        (they were sliced with the kept rows), down keeps the module's own bias."""
        comps = self.get_mlp_components(layer_idx, expert_idx=expert_idx)
        keep = (lambda m: m) if bias else (lambda m: None)
        up = self._swap_in(up_weights, keep(comps.up_proj), new_bias_u)
        gate = None if gate_weights is None else self._swap_in(gate_weights, keep(comps.gate_proj), new_bias_g)
        down_bias = None if comps.down_proj.bias is None else comps.down_proj.bias.data
        down = self._swap_in(down_weights, keep(comps.down_proj), down_bias)
        self.replace_mlp_layers(layer_idx, up, down, gate, expert_idx=expert_idx)

    @torch.no_grad()
    def slice_qk_dims(self, layer_idx: int, new_heads_Q: List[Tensor], new_heads_K: List[Tensor],
                      new_bias_Q: List[Tensor] = [], new_bias_K: List[Tensor] = [], bias: bool = True):
        """Per-head row blocks -> one q_proj / k_proj; a bias survives only if the original had one, `bias` is set and
        per-head biases were passed."""
        comps = self.get_attn_components(layer_idx)
        bq = torch.cat(list(new_bias_Q), dim=0) if len(new_bias_Q) > 0 else None
        bk = torch.cat(list(new_bias_K), dim=0) if len(new_bias_K) > 0 else None
        q = self._swap_in(torch.cat(list(new_heads_Q), dim=0), comps.q_proj if (bias and bq is not None) else None, bq)
        k = self._swap_in(torch.cat(list(new_heads_K), dim=0), comps.k_proj if (bias and bk is not None) else None, bk)
        self.replace_attn_layers(layer_idx, new_q=q, new_k=k, new_v=None, new_o=None)

    @torch.no_grad()
    def slice_vo_dims(self, layer_idx: int, new_heads_V: List[Tensor], new_heads_O: List[Tensor], bias: bool):
        """Per-head V row blocks / O column blocks -> v_proj (never biased) and o_proj (keeps the original bias)."""
        comps = self.get_attn_components(layer_idx)
        v = self._swap_in(torch.cat(list(new_heads_V), dim=0))
        o_bias = None if comps.o_proj.bias is None else comps.o_proj.bias.data
        o = self._swap_in(torch.cat(list(new_heads_O), dim=1), comps.o_proj if bias else None, o_bias)
        self.replace_attn_layers(layer_idx, new_q=None, new_k=None, new_v=v, new_o=o)

    # ---- shape properties (model_adapter.py:253-307) ----
    @property
    def arch(self) -> str:
        return self.model_config.model_type

    @property
    def n_layers(self) -> int:
        c = self.model_config
        return getattr(c, "n_layer", None) or getattr(c, "num_hidden_layers", None) or getattr(c, "num_layers", None)

    @property
    def n_heads(self) -> int:
        c = self.model_config
        return getattr(c, "n_head", None) or getattr(c, "num_attention_heads", None)

    @property
    def d_model(self) -> int:
        c = self.model_config
        return getattr(c, "hidden_size", None) or getattr(c, "dim", None)

    @property
    def d_int(self) -> int:
        return getattr(self.model_config, "intermediate_size", None)

    @property
    def head_dim(self) -> int:
        hd = getattr(self.model_config, "head_dim", None)
        return hd if hd else self.d_model // self.n_heads  # OPTConfig carries no head_dim

    @property
    def n_experts(self) -> int:
        if self.arch == "deepseek":
            return getattr(self.model_config, "n_routed_experts", 0)
        return getattr(self.model_config, "num_local_experts", 0)

    @property
    def n_kv_heads(self) -> int:
        return getattr(self.model_config, "num_key_value_heads", self.n_heads)

    def get_n_inner(self) -> int:
        return self.model_config.intermediate_size

    # ---- what an adapter must provide (model_adapter.py:249-392) ----
    @abstractmethod
    def compute_layer_energy(self, layer_idx: int, Ca: Optional[Tensor] = None) -> MLPTensors: ...

    @abstractmethod
    def calibrate_model(self, n_samples: int, batch_size: int, target_layers: List[int], dataset="wikitext"): ...

    @abstractmethod
    def get_transformer_blocks(self) -> nn.ModuleList: ...

    @abstractmethod
    def register_hooks(self, layer_idx: int, block: nn.Module, cov_mlp_list: List[Tensor], cov_q_list: List[Tensor],
                       cov_k_list: List[Tensor], cov_x_list: List[Tensor], handles: List[Any],
                       logger: logging.Logger): ...

    def on_batch_end_step(self, layer_idx: int, x_in: Tensor, cov_x_list: List[Tensor]):
        pass

    @abstractmethod
    def get_mlp_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> MLPComponents: ...

    @abstractmethod
    def get_mlp_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> MLPTensors: ...

    @abstractmethod
    def get_vo_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> VOComponents: ...

    @abstractmethod
    def get_vo_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> VOTensors: ...

    @abstractmethod
    def get_qk_components(self, layer_idx: int, expert_idx: Optional[int] = None) -> QKComponents: ...

    @abstractmethod
    def get_qk_tensors(self, layer_idx: int, expert_idx: Optional[int] = None) -> QKTensors: ...

    @abstractmethod
    def replace_mlp_layers(self, layer_idx: int, new_up: nn.Module, new_down: nn.Module,
                           new_gate: Optional[nn.Module] = None, expert_idx: Optional[int] = None) -> None: ...

    @abstractmethod
    def get_attn_components(self, layer_idx: int) -> AttentionComponents: ...

    @abstractmethod
    def replace_attn_layers(self, layer_idx: int, new_q: Optional[nn.Module], new_k: Optional[nn.Module],
                            new_v: Optional[nn.Module], new_o: Optional[nn.Module]) -> None: ...

    @abstractmethod
    def get_qk_weights(self, layer_idx: int) -> Tuple[Tensor, Tensor]: ...

    @abstractmethod
    def get_vo_weights(self, layer_idx: int) -> Tuple[Tensor, Tensor]: ...

    # ---- statistic hooks shared by adapters (model_adapter.py:546-567) ----
    @staticmethod
    def _make_fc_hook(layer_idx, cov_mlp_list):
        """sigma_mlp += ReLU(fc1 out)^T ReLU(fc1 out) -- ReLU fused into the kernel's load."""
        @torch.no_grad()
        def hook(module, inp, out):
            ops.cov_accum(cov_mlp_list[layer_idx], out, relu=True)
        return hook

    @staticmethod
    def _make_proj_hook(layer_idx, cov_list, n_heads, head_dim, d_model=None):
        """sigma[h] += P_h^T P_h for every head, read in place from the [tokens, n_heads*head_dim] output."""
        @torch.no_grad()
        def hook(module, inp, out):
            ops.cov_accum(cov_list[layer_idx], out, n_heads=n_heads)
        return hook
