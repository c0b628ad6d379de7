"""Attention of a compressed model.

A compressed layer keeps r_qk <= head_dim columns of every q/k head (chosen per kv head, in RoPE pairs, order given by
the layer's rotary mask) and r_vo columns of every v/o head.  The stock HF attention modules assume `head_dim`
everywhere, so the reference ships forked copies of the HF modeling files with the checkpoint
(src/patchers/*Rebuild.py, loaded through config.auto_map).  This engine does not fork the modeling files: it keeps the
stock HF modules and gives their attention the compressed semantics, with the elementwise chain in one HIP kernel:

  * q/k/v are viewed with their own per-layer head widths                        (LlamaRebuild.py:320-326)
  * RoPE: cos/sin are gathered along the feature axis by the rotary mask, per kv head, query heads of a group share
    their kv head's mask; rotate_half pairs the two halves of the KEPT columns   (LlamaRebuild.py:153-176)
    -> modegpt_amd.ops.rope_gather (csrc/rope.hip), which also writes the [B, heads, T, r] layout attention reads
  * Qwen3: q_norm / k_norm normalise over the kept columns with the norm weight gathered by the same mask
    (DenseQwenRebuild.py:262-286) -> fused into the same kernel
  * softmax scale = (compressed q/k head width) ** -0.5                          (LlamaRebuild.py:266,282)
  * OPT: no RoPE, no mask; q is pre-scaled by the compressed width, attention runs with scale 1  (OPTRebuild.py:144-163)
  * the attention product itself goes through HF's attention interface exactly as in the stock modules
    (config._attn_implementation: sdpa / eager / ...) -- PyTorch plumbing, not part of this engine

Two entry points share the forward functions:
  install_compressed_attention(adapter, rotary_masks)   live model right after ModelAdapter.convert_model
  shrink_to_config_ranks(model, arch)                   called by the modeling files shipped with a checkpoint
                                                         (patchers/{LlamaRebuild,DenseQwenRebuild,OPTRebuild}.py)

This file travels WITH a compressed checkpoint (model_utils.save_compressed_model copies it next to the *Rebuild.py that
imports it relatively), the way the reference ships its self-contained modeling file (src/model_utils.py:103-124): it
depends on torch and transformers only.  The fused HIP kernel is used when the modegpt_amd package is importable and the
tensors live on a GPU; anywhere else (another machine, a CPU) the same chain runs as plain torch ops, the reference's own
expression op for op -- the checkpoint loads and evaluates wherever the reference's would.  Which path ran is counted in
PATH_CALLS; MODEGPT_REQUIRE_HIP=1 turns the torch path into an error.  One case raises by itself: the engine is installed here but
its library does not load (a broken installation) and the tensors are on a GPU -- unless MODEGPT_ALLOW_TORCH=1.

Inference only: the kernel has no backward.
"""
from __future__ import annotations

import logging
import os
import types
from typing import List, Optional

import torch
import torch.nn as nn

logger = logging.getLogger("MoDeGPT")
PATH_CALLS = {"hip": 0, "torch": 0}     # rope/norm chains served by mdg_rope_gather / by the torch expression
_HIP_OPS = None
_HIP_BROKEN = None                      # why the engine, although installed here, cannot serve (its library failed to load)


def _hip_ops():
    """modegpt_amd.ops when this process can import the engine AND its library loads, else None -- decided once.  Two different
    reasons for None: the package is simply not on this machine (a checkpoint taken elsewhere: the portable torch path, with one
    warning), or it is here and libmodegpt_hip.so does not load (a broken installation: _HIP_BROKEN holds the reason and GPU
    tensors raise instead of quietly running ten eager passes per layer)."""
    global _HIP_OPS, _HIP_BROKEN
    if _HIP_OPS is None:
        try:
            import modegpt_amd  # noqa: F401
        except Exception as exc:  # not installed here: the portable path
            _HIP_OPS = False
            logger.warning("compressed attention: modegpt_amd is not importable (%s); running the reference's torch expression", exc)
            return None
        try:
            from modegpt_amd import _lib, ops as _ops
            if torch.cuda.is_available():
                _lib.load()
            _HIP_OPS = _ops
        except Exception as exc:
            _HIP_OPS = False
            _HIP_BROKEN = f"{type(exc).__name__}: {exc}"
    return _HIP_OPS or None


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _rope_gather_torch(x, cos, sin, mask, n_heads, n_kv, norm_weight=None, eps=1e-6):
    """The reference's chain in torch (LlamaRebuild.py:153-186, DenseQwenRebuild.py:262-286): x [B, T, n_heads * r] ->
    optional masked RMSNorm over the r kept columns (fp32, weight gathered by the mask) -> [B, n_heads, T, r] ->
    x * cos[mask] + rotate_half(x) * sin[mask], every op rounded to the tensor dtype as eager torch does."""
    B, T, width = x.shape
    r = width // n_heads
    x = x.view(B, T, n_heads, r)
    m = None
    if mask is not None:
        m = mask.to(x.device)
        if n_heads != n_kv:
            m = torch.repeat_interleave(m, n_heads // n_kv, dim=0)         # query heads share their kv head's row
    if norm_weight is not None:
        xf = x.to(torch.float32)
        normed = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
        x = (norm_weight[m][None, None] * normed).to(x.dtype)
    x = x.transpose(1, 2)                                                  # [B, n_heads, T, r]
    if m is None:
        c, s = cos.unsqueeze(1), sin.unsqueeze(1)
    else:                                                                  # [1, T, H, r] -> [1, H, T, r]
        c, s = cos[:1, :, m].permute(0, 2, 1, 3), sin[:1, :, m].permute(0, 2, 1, 3)
    return x * c + _rotate_half(x) * s


def _rope_gather(x, cos, sin, mask, n_heads, n_kv, head_dim, norm_weight=None, eps=1e-6):
    hip = _hip_ops() if x.is_cuda else None
    if hip is not None:
        PATH_CALLS["hip"] += 1
        return hip.rope_gather(x, cos, sin, mask, n_heads, n_kv, head_dim, norm_weight=norm_weight, eps=eps)
    if os.environ.get("MODEGPT_REQUIRE_HIP", "0") == "1":
        raise RuntimeError("compressed attention: MODEGPT_REQUIRE_HIP=1 but the HIP kernel cannot serve this call "
                           f"(tensor on {x.device}, modegpt_amd importable: {_hip_ops() is not None})")
    if x.is_cuda and _HIP_BROKEN and os.environ.get("MODEGPT_ALLOW_TORCH", "0") != "1":
        # the engine IS installed here and its library does not load: a broken installation on a GPU box, not portability -- say so.
        # (A machine without the engine -- a CUDA box, lm-eval -- takes the portable torch path below with the one warning of
        # _hip_ops, as the reference's self-contained checkpoints do; MODEGPT_REQUIRE_HIP=1 is the strict opt-in.)
        raise RuntimeError("compressed attention: modegpt_amd is installed but libmodegpt_hip.so cannot be used in this process "
                           f"({_HIP_BROKEN}), so mdg_rope_gather is unavailable for these GPU tensors.  Rebuild the engine, or set "
                           "MODEGPT_ALLOW_TORCH=1 to run the reference's torch expression on the GPU instead.")
    PATH_CALLS["torch"] += 1
    return _rope_gather_torch(x, cos.to(x.dtype), sin.to(x.dtype), mask, n_heads, n_kv,
                              None if norm_weight is None else norm_weight.detach().to(x.dtype), eps)


def _attention_interface(module):
    """HF's dispatch on config._attn_implementation, falling back to the model family's eager function."""
    import importlib

    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    eager = importlib.import_module(type(module).__module__).eager_attention_forward
    impl = getattr(module.config, "_attn_implementation", "eager") or "eager"
    if hasattr(ALL_ATTENTION_FUNCTIONS, "get_interface"):
        return ALL_ATTENTION_FUNCTIONS.get_interface(impl, eager)
    return eager if impl == "eager" else ALL_ATTENTION_FUNCTIONS[impl]


def _rope_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
    """Llama / Qwen3 (RoPE, optional GQA, optional per-head q/k RMSNorm)."""
    if torch.is_grad_enabled() and hidden_states.requires_grad:
        raise RuntimeError("compressed attention is inference-only (mdg_rope_gather has no backward); use torch.no_grad()")
    B, T, _ = hidden_states.shape
    cfg = self.config
    n_h, n_kv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, self.head_dim
    q = self.q_proj(hidden_states)
    k = self.k_proj(hidden_states)
    v = self.v_proj(hidden_states)
    r_qk = q.shape[-1] // n_h
    v = v.view(B, T, n_kv, v.shape[-1] // n_kv).transpose(1, 2)
    cos, sin = position_embeddings                               # [B or 1, T, head_dim]
    mask = self.layer_rotary_mask                                # int64 [n_kv, r_qk] or None
    if mask is not None and mask.device != q.device:             # a plain attribute (see _install): moved on first use
        mask = self.layer_rotary_mask = mask.to(q.device)
    qw = kw = None
    eps = 1e-6
    if getattr(self, "q_norm", None) is not None:
        if mask is None:                                         # QK stage not run: the stock per-head norm
            q = self.q_norm(q.view(B, T, n_h, r_qk)).view(B, T, -1)
            k = self.k_norm(k.view(B, T, n_kv, r_qk)).view(B, T, -1)
        else:
            qw, kw, eps = self.q_norm.weight, self.k_norm.weight, self.q_norm.variance_epsilon
    if mask is not None:
        # the reference's gather index has batch extent 1 (LlamaRebuild.py:167-175): batch 0's table serves every batch
        cos, sin = cos[:1], sin[:1]
    q = _rope_gather(q, cos, sin, mask, n_h, n_kv, hd, norm_weight=qw, eps=eps)         # [B, n_h,  T, r_qk]
    k = _rope_gather(k, cos, sin, mask, n_kv, n_kv, hd, norm_weight=kw, eps=eps)        # [B, n_kv, T, r_qk]
    if past_key_values is not None:
        k, v = past_key_values.update(k, v, self.layer_idx)
    extra = {"sliding_window": self.sliding_window} if getattr(self, "sliding_window", None) is not None else {}
    out, weights = _attention_interface(self)(self, q, k, v, attention_mask, dropout=0.0, scaling=float(r_qk) ** -0.5,
                                              **extra, **kwargs)
    return self.o_proj(out.reshape(B, T, -1).contiguous()), weights


def _opt_forward(self, hidden_states, past_key_values=None, attention_mask=None, output_attentions=False, **kwargs):
    B, T, _ = hidden_states.shape
    n_h = self.num_heads
    q = self.q_proj(hidden_states)
    r_qk = q.shape[-1] // n_h
    q = q * (float(r_qk) ** -0.5)                                # pre-scaled, as upstream OPT does; scale 1 below
    k = self.k_proj(hidden_states)
    v = self.v_proj(hidden_states)
    q = q.view(B, T, n_h, r_qk).transpose(1, 2)
    k = k.view(B, T, n_h, r_qk).transpose(1, 2)
    v = v.view(B, T, n_h, v.shape[-1] // n_h).transpose(1, 2)
    if past_key_values is not None:
        k, v = past_key_values.update(k, v, self.layer_idx)
    out, weights = _attention_interface(self)(self, q, k, v, attention_mask, dropout=0.0, scaling=1.0, **kwargs)
    return self.out_proj(out.reshape(B, T, -1).contiguous()), (weights if output_attentions else None)


def _check_mask(mask: torch.Tensor, n_kv: int, head_dim: int, layer: int) -> torch.Tensor:
    """The kernel clamps indices for memory safety only; a mask that does not index head_dim is refused here, once."""
    if mask.dim() != 2 or mask.shape[0] != n_kv or mask.shape[1] % 2:
        raise ValueError(f"layer {layer}: rotary mask must be [n_kv={n_kv}, even width], got {tuple(mask.shape)}")
    mask = mask.to(torch.int64)
    if mask.numel() and (int(mask.min()) < 0 or int(mask.max()) >= head_dim):
        raise ValueError(f"layer {layer}: rotary mask indexes outside head_dim={head_dim}")
    return mask.contiguous()


def _attention_modules(model, arch: str):
    blocks = model.model.decoder.layers if arch == "opt" else model.model.layers
    return [b.self_attn for b in blocks]


def _install(model, arch: str, rotary_masks: Optional[List[torch.Tensor]]) -> None:
    attns = _attention_modules(model, arch)
    if rotary_masks is not None and len(rotary_masks) != len(attns):
        raise ValueError(f"{len(rotary_masks)} rotary masks for {len(attns)} layers")
    for i, attn in enumerate(attns):
        if arch == "opt":
            attn.forward = types.MethodType(_opt_forward, attn)
            continue
        mask = None
        if rotary_masks is not None and rotary_masks[i] is not None:
            mask = _check_mask(rotary_masks[i], attn.config.num_key_value_heads, attn.head_dim, i)
        # a plain attribute, as upstream (LlamaRebuild.py:310): HF materialises unknown buffers of a model built on the
        # meta device as uninitialised memory, which would silently replace the mask
        attn.layer_rotary_mask = mask
        attn.forward = types.MethodType(_rope_forward, attn)


def install_compressed_attention(adapter, rotary_masks: Optional[List[torch.Tensor]]) -> None:
    """Give every attention module of `adapter.model` the compressed-aware forward.  `rotary_masks`: the list
    `compress_qk` returned (one int64 [n_kv, r_qk] tensor per layer, layer order) or None for architectures without
    RoPE masks (OPT) / for an uncompressed QK stage."""
    _install(adapter.model, "opt" if adapter.arch == "opt" else adapter.arch, rotary_masks)


def _resize(linear: nn.Linear, out_features: Optional[int] = None, in_features: Optional[int] = None) -> nn.Linear:
    """A bias-free Linear of the compressed shape (ModelAdapter.convert_model builds bias-free modules,
    model_adapter.py:199-208, so the checkpoint holds no bias for them), on the device / dtype of the one it replaces."""
    o = linear.out_features if out_features is None else out_features
    i = linear.in_features if in_features is None else in_features
    if (o, i) == (linear.out_features, linear.in_features) and linear.bias is None:
        return linear
    return nn.Linear(i, o, bias=False, device=linear.weight.device, dtype=linear.weight.dtype)


def shrink_to_config_ranks(model, arch: str) -> None:
    """Called from the constructor of a modeling file shipped with a compressed checkpoint, BEFORE the weights are loaded:
    give every projection the per-layer shape recorded by patch_config (q_ranks / k_ranks / v_ranks / o_ranks /
    gate_ranks; qk_ranks / vo_ranks for OPT), read rotary_masks.pt from config.mask_path, install the forward."""
    cfg = model.config
    attns = _attention_modules(model, arch)
    blocks = model.model.decoder.layers if arch == "opt" else model.model.layers
    for i, (block, attn) in enumerate(zip(blocks, attns)):
        if arch == "opt":
            qk, vo = cfg.qk_ranks[i], cfg.vo_ranks[i]
            attn.q_proj, attn.k_proj = _resize(attn.q_proj, qk), _resize(attn.k_proj, qk)
            attn.v_proj, attn.out_proj = _resize(attn.v_proj, vo), _resize(attn.out_proj, in_features=vo)
            block.fc1, block.fc2 = _resize(block.fc1, cfg.gate_ranks[i]), _resize(block.fc2, in_features=cfg.gate_ranks[i])
        else:
            attn.q_proj, attn.k_proj = _resize(attn.q_proj, cfg.q_ranks[i]), _resize(attn.k_proj, cfg.k_ranks[i])
            attn.v_proj, attn.o_proj = _resize(attn.v_proj, cfg.v_ranks[i]), _resize(attn.o_proj, in_features=cfg.o_ranks[i])
            g = cfg.gate_ranks[i]
            block.mlp.gate_proj, block.mlp.up_proj = _resize(block.mlp.gate_proj, g), _resize(block.mlp.up_proj, g)
            block.mlp.down_proj = _resize(block.mlp.down_proj, in_features=g)
    masks = None
    path = getattr(cfg, "mask_path", None)
    if arch != "opt" and path:
        if not os.path.exists(path):
            raise FileNotFoundError(f"config.mask_path = {path} does not exist (the checkpoint stores an absolute path, "
                                    "as upstream does: model_utils.py:105-112)")
        masks = torch.load(path, map_location="cpu")
    _install(model, arch, masks)
