"""Generate tests/golden/ckpt_{llama,qwen3,opt}.npz: a tiny compressed checkpoint written by THIS engine's writer, loaded and
run through the REFERENCE's modeling files (src/patchers/{LlamaRebuild,DenseQwenRebuild,OPTRebuild}.py) on the CPU.

TEST INFRASTRUCTURE ONLY, build container only (/root/reference does not exist on the GPU box; the vectors travel).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_checkpoint_golden.py [--ref /root/reference]

What this pins ("so the existing patchers load the build's output unchanged"):
  1. the artefact set of modegpt_amd.model_utils.save_compressed_model (+ ModelAdapter.patch_config) -- weights file, config
     fields q_ranks / k_ranks / v_ranks / o_ranks / gate_ranks / ffn_dim / mask_path / auto_map, rotary_masks.pt -- is what the
     reference's loader expects: the reference's own *Rebuild.py is dropped into the checkpoint directory in place of the
     shipped one (temporary directory only; no reference source is copied into the repository) and
     AutoModelForCausalLM.from_pretrained(dir, trust_remote_code=True) must build the model and load every tensor;
  2. the logits of that reference-modelled forward on fixed token ids are stored; tests/ compare the logits of the SAME
     checkpoint loaded through this engine's shipped modeling file (torch path on the CPU, HIP kernel on the GPU) with them.
The reference's LlamaModel / Qwen3Model read the masks with torch.load(mask_path, map_location="cuda") (LlamaRebuild.py:449):
for the duration of the load torch.load's map_location "cuda" is redirected to "cpu".
The compressed weights are random stand-ins with per-layer ranks that differ between layers and valid RoPE-pair masks (the
compression arithmetic is pinned elsewhere; this fixture is about the checkpoint format and the compressed forward).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

KINDS = {"llama": "LlamaRebuild.py", "qwen3": "DenseQwenRebuild.py", "opt": "OPTRebuild.py"}
VOCAB, T_MAX = 97, 32


def bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16) if t.dtype == torch.bfloat16 else t.detach().numpy()


def tiny_model(kind: str):
    import transformers
    torch.manual_seed({"llama": 1, "qwen3": 2, "opt": 3}[kind])
    if kind == "opt":
        cfg = transformers.OPTConfig(hidden_size=64, ffn_dim=160, num_hidden_layers=2, num_attention_heads=4,
                                     vocab_size=VOCAB, max_position_embeddings=T_MAX, word_embed_proj_dim=64)
        return transformers.OPTForCausalLM(cfg).to(torch.bfloat16).eval()
    common = dict(hidden_size=64, intermediate_size=160, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                  head_dim=16, vocab_size=VOCAB, max_position_embeddings=T_MAX)
    if kind == "qwen3":
        return transformers.Qwen3ForCausalLM(transformers.Qwen3Config(**common)).to(torch.bfloat16).eval()
    return transformers.LlamaForCausalLM(transformers.LlamaConfig(**common)).to(torch.bfloat16).eval()


def compress_stand_in(ad, kind: str):
    """Per-layer compressed shapes (different in every layer) with random bf16 weights; rotary masks of the shape compress_qk
    produces (score-ordered pair indices, then the same + head_dim / 2)."""
    g = torch.Generator().manual_seed(17)
    n_h, n_kv, hd, d = ad.n_heads, ad.n_kv_heads, ad.head_dim, ad.d_model
    masks = []

    def lin(rows, cols):
        m = torch.nn.Linear(cols, rows, bias=False, dtype=torch.bfloat16)
        m.weight.data.copy_((torch.randn(rows, cols, generator=g) * (cols ** -0.5)).to(torch.bfloat16))
        return m

    for i in range(ad.n_layers):
        r_qk, r_mlp = 8 + 2 * i, 100 + 7 * i
        # Llama / Qwen3: the same (even) head width for q/k and v/o, as compress_qk.py:176-182 and compress_vo.py:36-41 produce
        # for one keep ratio -- the reference's attention views q, k AND v with q_ranks // n_heads (LlamaRebuild.py:266,326)
        r_vo = 9 + i if kind == "opt" else r_qk
        ad.replace_attn_layers(i, lin(n_h * r_qk, d), lin(n_kv * r_qk, d), lin(n_kv * r_vo, d), lin(d, n_h * r_vo))
        ad.replace_mlp_layers(i, lin(r_mlp, d), lin(d, r_mlp), None if kind == "opt" else lin(r_mlp, d))
        idx = torch.stack([torch.randperm(hd // 2, generator=g)[:r_qk // 2] for _ in range(n_kv)])
        masks.append(torch.cat((idx, idx + hd // 2), dim=1))
    return masks


@contextlib.contextmanager
def cuda_means_cpu_for_load():
    real = torch.load

    def load(f, *a, **kw):
        ml = kw.get("map_location")
        if (isinstance(ml, str) and ml.startswith("cuda")) or (isinstance(ml, torch.device) and ml.type == "cuda"):
            kw["map_location"] = "cpu"
        return real(f, *a, **kw)

    torch.load = load
    try:
        yield
    finally:
        torch.load = real


@contextlib.contextmanager
def transformers_4x_names():
    """The reference's modeling files were written against transformers 4.5x; this image has 5.x, where two names they look up
    moved.  For the duration of the load the LIBRARY'S OWN objects are registered under the old names -- nothing of the
    reference is edited, nothing is re-implemented:
      * ROPE_INIT_FUNCTIONS["default"] (LlamaRebuild.py:86) -> LlamaRotaryEmbedding.compute_default_rope_parameters, the
        function 5.x calls for the default rope type instead of a registry entry."""
    from transformers import modeling_rope_utils as mru
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    added = "default" not in mru.ROPE_INIT_FUNCTIONS
    if added:
        mru.ROPE_INIT_FUNCTIONS["default"] = LlamaRotaryEmbedding.compute_default_rope_parameters
    try:
        yield ["ROPE_INIT_FUNCTIONS['default'] -> LlamaRotaryEmbedding.compute_default_rope_parameters"] if added else []
    finally:
        if added:
            del mru.ROPE_INIT_FUNCTIONS["default"]


class _OptDone(Exception):
    """(control flow: the OPT branch has produced its logits inside the load context)"""


def forward_logits(model, ids):
    with torch.no_grad():
        return model(input_ids=ids).logits.float()


def forward_through_layers(model, ids):
    """Logits of a Llama / Qwen3 style model by driving its OWN modules in order -- embedding, rotary table, every decoder layer
    with an explicit additive causal mask, final norm, lm_head.  Used for the reference's classes, whose model-level forward
    calls transformers' mask builder with a 4.x keyword (create_causal_mask(input_embeds=...)); everything specific to a
    compressed checkpoint -- per-layer widths, the rotary-mask gather, the softmax scale -- lives in the layers driven here."""
    core = model.model
    with torch.no_grad():
        h = core.embed_tokens(ids)
        B, T = ids.shape
        pos = torch.arange(T)[None].expand(B, T)
        cos_sin = core.rotary_emb(h, pos)
        mask = torch.full((T, T), torch.finfo(h.dtype).min, dtype=h.dtype).triu(1)[None, None]
        for layer in core.layers:
            out = layer(h, attention_mask=mask, position_ids=pos, position_embeddings=cos_sin)
            h = out[0] if isinstance(out, tuple) else out
        return model.lm_head(core.norm(h)).float()


def opt_reference_logits(cls, rconf, refdir, ids, state):
    """The reference's OPT modules driven on this writer's checkpoint.  OPTRebuild.OPTForCausalLM does not construct under
    transformers 5 (its `_tied_weights_keys` is a 4.x-style list), so -- as for Llama / Qwen3 around their model-level forward --
    the part that carries the compressed semantics is driven directly: the reference's OPTModel (OPTRebuild.py:371, 745) is
    constructed from our config (per-layer qk_ranks / vo_ranks / gate_ranks), filled with our tensors, and its modules run in the
    order its own OPTDecoder.forward runs them (OPTRebuild.py:560-760): embed_tokens + embed_positions(attention_mask, 0,
    position_ids), every OPTDecoderLayer with an explicit additive causal mask, final_layer_norm; the tied lm_head of
    OPTForCausalLM (OPTRebuild.py:809-813, 905) is the product with embed_tokens.weight.  The checkpoint carries no bias for
    the six compressed Linears of a layer (model_adapter.py:199-208 rebuilds them bias-free) while OPTRebuild creates them with
    config.enable_bias: those 6 x n_layers keys are the only ones missing, and they are zeroed -- what the reference's own
    from_pretrained does with them (OPTPreTrainedModel._init_weights, OPTRebuild.py:356-361)."""
    import sys

    from safetensors.torch import load_file
    mod = sys.modules[cls.__module__]
    with torch.device("meta"):
        ref = mod.OPTModel(rconf)
    ref = ref.to_empty(device="cpu").to(torch.bfloat16).eval()
    sd = {}
    for f in os.listdir(refdir):
        if f.endswith(".safetensors"):
            sd.update(load_file(os.path.join(refdir, f)))
        elif f.endswith(".bin"):
            sd.update(torch.load(os.path.join(refdir, f), map_location="cpu"))
    sub = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    res = ref.load_state_dict(sub, strict=False)
    compressed = ("self_attn.q_proj.bias", "self_attn.k_proj.bias", "self_attn.v_proj.bias", "self_attn.out_proj.bias", "fc1.bias", "fc2.bias")
    assert not res.unexpected_keys, res.unexpected_keys[:4]
    assert all(k.endswith(compressed) for k in res.missing_keys) and len(res.missing_keys) == 6 * rconf.num_hidden_layers, res.missing_keys[:8]
    own = dict(ref.named_parameters())
    for k in res.missing_keys:
        own[k].data.zero_()
    rs = ref.state_dict()
    for k, v in state.items():                     # every tensor of our checkpoint sits in the reference's module, bit for bit
        kk = k[len("model."):] if k.startswith("model.") else None
        if kk is not None:
            assert rs[kk].shape == v.shape and torch.equal(rs[kk], v), k
    ref.config._attn_implementation = "eager"
    dec = ref.decoder
    with torch.no_grad():
        B, T = ids.shape
        am = torch.ones(B, T, dtype=torch.long)
        pos = (torch.cumsum(am, dim=1) * am - 1).long()
        h = dec.embed_tokens(ids)
        if dec.project_in is not None:
            h = dec.project_in(h)
        h = h + dec.embed_positions(am, 0, position_ids=pos)
        mask = torch.full((T, T), torch.finfo(h.dtype).min, dtype=h.dtype).triu(1)[None, None]
        for layer in dec.layers:
            out = layer(h, attention_mask=mask)
            h = out[0] if isinstance(out, tuple) else out
        if dec.final_layer_norm is not None:
            h = dec.final_layer_norm(h)
        if dec.project_out is not None:
            h = dec.project_out(h)
        return torch.nn.functional.linear(h, dec.embed_tokens.weight).float(), ref


def opt_drive_own(model, ids):
    """The same drive through THIS engine's shipped modeling file (stock OPT modules with the compressed forward installed)."""
    dec = model.model.decoder
    with torch.no_grad():
        B, T = ids.shape
        am = torch.ones(B, T, dtype=torch.long)
        pos = (torch.cumsum(am, dim=1) * am - 1).long()
        h = dec.embed_tokens(ids)
        if dec.project_in is not None:
            h = dec.project_in(h)
        h = h + dec.embed_positions(am, 0, position_ids=pos)
        mask = torch.full((T, T), torch.finfo(h.dtype).min, dtype=h.dtype).triu(1)[None, None]
        for layer in dec.layers:
            out = layer(h, attention_mask=mask)
            h = out[0] if isinstance(out, tuple) else out
        if dec.final_layer_norm is not None:
            h = dec.final_layer_norm(h)
        if dec.project_out is not None:
            h = dec.project_out(h)
        return model.lm_head(h).float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--kinds", default="llama,qwen3,opt")
    a = ap.parse_args()
    import transformers
    from modegpt_amd.adapters.model_adapter import ModelAdapter
    from modegpt_amd.model_utils import save_compressed_model
    for kind in a.kinds.split(","):
        model = tiny_model(kind)
        ad = ModelAdapter.from_model(model, None)
        masks = compress_stand_in(ad, kind)
        ad.patch_config()
        ids = torch.randint(0, VOCAB, (2, 12), generator=torch.Generator().manual_seed(5))
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "model")
            save_compressed_model(ad, rotary_masks=masks if kind != "opt" else [], save_dir=out, source_model_name="none")
            cfg_json = json.load(open(os.path.join(out, "config.json")))
            state = {k: v.clone() for k, v in model.state_dict().items()}
            # (A) this engine's shipped modeling file, torch path on the CPU
            mine = transformers.AutoModelForCausalLM.from_pretrained(out, trust_remote_code=True, dtype=torch.bfloat16).eval()
            logits_mine = forward_logits(mine, ids)
            logits_mine_layers = None
            # (B) the reference's modeling file in its place
            ref_file = os.path.join(a.ref, "src", "patchers", KINDS[kind])
            shutil.copy(ref_file, os.path.join(out, KINDS[kind]))
            os.remove(os.path.join(out, "compressed_attention.py"))
            status = "loaded"
            try:
                with cuda_means_cpu_for_load(), transformers_4x_names() as shims:
                    # a distinct module name per load: HF caches dynamic modules by directory name
                    refdir = os.path.join(tmp, "model_ref")
                    shutil.copytree(out, refdir)
                    refcfg = json.load(open(os.path.join(refdir, "config.json")))
                    if refcfg.get("mask_path"):
                        refcfg["mask_path"] = os.path.join(refdir, "rotary_masks.pt")
                        json.dump(refcfg, open(os.path.join(refdir, "config.json"), "w"))
                    # The class config.auto_map names, resolved the way from_pretrained resolves it; then constructed and
                    # filled directly.  (from_pretrained itself goes on to re-initialise "missing" non-persistent buffers
                    # through transformers-5 hooks -- module.compute_default_rope_parameters -- that a 4.x-era modeling file
                    # does not have; constructing the class and loading the state dict is the same model without that step.)
                    from transformers.dynamic_module_utils import get_class_from_dynamic_module
                    rconf = transformers.AutoConfig.from_pretrained(refdir, trust_remote_code=True)
                    cls = get_class_from_dynamic_module(rconf.auto_map["AutoModelForCausalLM"], refdir)
                    if kind == "opt":
                        mine.config._attn_implementation = "eager"
                        logits_ref, ref = opt_reference_logits(cls, rconf, refdir, ids, state)
                        logits_mine_layers = opt_drive_own(mine, ids)
                        shims = list(shims) + ["OPTForCausalLM does not construct under transformers 5: the reference's OPTModel is "
                                               "constructed, filled and driven module by module; the 6 bias keys per layer the "
                                               "checkpoint lacks are zeroed as the reference's _init_weights does"]
                        d2 = (logits_mine_layers - logits_mine).abs().max().item()
                        d3 = (logits_mine_layers - logits_ref).abs().max().item()
                        print(f"[opt] this engine's model: module-by-module drive (eager attention) vs its model-level forward (sdpa): "
                              f"max |diff| = {d2:.3e};  module-by-module drive, this engine's modules vs the reference's: {d3:.3e}")
                        raise _OptDone()
                    # built on the meta device (transformers 5 runs no weight initialisers there -- its initialiser for
                    # rotary modules calls a 5.x-only method), materialised empty, filled from the checkpoint; the rotary
                    # module, whose inv_freq is a computed non-persistent buffer, is constructed again on the CPU
                    with torch.device("meta"):
                        ref = cls(rconf)
                    ref = ref.to_empty(device="cpu").to(torch.bfloat16).eval()
                    core = ref.model.decoder if kind == "opt" else ref.model
                    if hasattr(core, "rotary_emb"):
                        core.rotary_emb = type(core.rotary_emb)(rconf)
                    from safetensors.torch import load_file
                    wfile = [f for f in os.listdir(refdir) if f.endswith(".safetensors") or f.endswith(".bin")]
                    sd = {}
                    for f in wfile:
                        sd.update(load_file(os.path.join(refdir, f)) if f.endswith(".safetensors")
                                  else torch.load(os.path.join(refdir, f), map_location="cpu"))
                    res = ref.load_state_dict(sd, strict=False)
                    unexpected = list(res.unexpected_keys)
                    missing_persistent = [k for k in res.missing_keys if "lm_head" not in k]   # lm_head is tied to the embedding
                    assert not unexpected and not missing_persistent, (unexpected[:4], missing_persistent[:4])
                    if hasattr(ref, "tie_weights"):
                        ref.tie_weights()
                assert type(ref).__module__.endswith(KINDS[kind][:-3])
                rs = ref.state_dict()
                missing = [k for k in state if k not in rs]
                assert not missing, f"reference model lacks {missing[:4]}"
                for k, v in state.items():
                    assert rs[k].shape == v.shape and torch.equal(rs[k], v), k
                ref.config._attn_implementation = "eager"
                if kind == "opt":
                    logits_ref = forward_logits(ref, ids)
                else:
                    logits_ref = forward_through_layers(ref, ids)
                    mine.config._attn_implementation = "eager"
                    logits_mine_layers = forward_through_layers(mine, ids)
                    d2 = (logits_mine_layers - logits_mine).abs().max().item()
                    d3 = (logits_mine_layers - logits_ref).abs().max().item()
                    print(f"[{kind}] this engine's model: layer-by-layer drive (eager attention) vs its model-level forward (sdpa): "
                          f"max |diff| = {d2:.3e};  layer-by-layer drive, this engine's modules vs the reference's: {d3:.3e}")
            except _OptDone:
                pass
            except Exception as exc:  # recorded, not hidden: the fixture says what the reference's file did with the checkpoint
                import traceback
                traceback.print_exc()
                status = f"reference modeling file failed: {type(exc).__name__}: {str(exc)[:300]}"
                logits_ref = None
            print(f"[{kind}] reference {KINDS[kind]}: {status}")
            if logits_ref is not None:
                d = (logits_mine - logits_ref).abs().max().item()
                print(f"[{kind}] logits through this engine's modeling file vs the reference's: max |diff| = {d:.3e} "
                      f"(max |logit| = {logits_ref.abs().max().item():.3f})")
        fx = {"meta_kind": np.array(kind), "meta_status": np.array(status), "meta_shims": np.array("; ".join(shims)),
              "meta_transformers": np.array(transformers.__version__), "config_json": np.array(json.dumps(cfg_json)),
              "input_ids": ids.numpy(), "n_masks": np.array(len(masks) if kind != "opt" else 0)}
        for i, m in enumerate(masks if kind != "opt" else []):
            fx[f"mask_{i}"] = m.numpy()
        for k, v in state.items():
            fx["w:" + k] = bits(v)
            fx["dtype:" + k] = np.array(str(v.dtype))
        if logits_ref is not None:
            fx["logits_reference"] = logits_ref.numpy()
        fx["logits_engine_torch_path"] = logits_mine.numpy()          # model-level forward (sdpa), this engine's modeling file, CPU
        if logits_mine_layers is not None:
            fx["logits_engine_layers"] = logits_mine_layers.numpy()  # layer-by-layer drive (eager attention), same file
        np.savez_compressed(os.path.join(a.out, f"ckpt_{kind}.npz"), **fx)


if __name__ == "__main__":
    main()
