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
        r_qk, r_vo, r_mlp = 8 + 2 * i, 9 + i, 100 + 7 * i
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


def forward_logits(model, ids):
    with torch.no_grad():
        return model(input_ids=ids).logits.float()


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
            # (B) the reference's modeling file in its place
            ref_file = os.path.join(a.ref, "src", "patchers", KINDS[kind])
            shutil.copy(ref_file, os.path.join(out, KINDS[kind]))
            os.remove(os.path.join(out, "compressed_attention.py"))
            status = "loaded"
            try:
                with cuda_means_cpu_for_load():
                    # a distinct module name per load: HF caches dynamic modules by directory name
                    refdir = os.path.join(tmp, "model_ref")
                    shutil.copytree(out, refdir)
                    refcfg = json.load(open(os.path.join(refdir, "config.json")))
                    if refcfg.get("mask_path"):
                        refcfg["mask_path"] = os.path.join(refdir, "rotary_masks.pt")
                        json.dump(refcfg, open(os.path.join(refdir, "config.json"), "w"))
                    ref = transformers.AutoModelForCausalLM.from_pretrained(refdir, trust_remote_code=True,
                                                                            dtype=torch.bfloat16).eval()
                assert type(ref).__module__.endswith(KINDS[kind][:-3])
                rs = ref.state_dict()
                missing = [k for k in state if k not in rs]
                assert not missing, f"reference model lacks {missing[:4]}"
                for k, v in state.items():
                    assert rs[k].shape == v.shape and torch.equal(rs[k], v), k
                logits_ref = forward_logits(ref, ids)
            except Exception as exc:  # recorded, not hidden: the fixture says what the reference's file did with the checkpoint
                status = f"reference modeling file failed: {type(exc).__name__}: {str(exc)[:300]}"
                logits_ref = None
            print(f"[{kind}] reference {KINDS[kind]}: {status}")
            if logits_ref is not None:
                d = (logits_mine - logits_ref).abs().max().item()
                print(f"[{kind}] logits through this engine's modeling file vs the reference's: max |diff| = {d:.3e} "
                      f"(max |logit| = {logits_ref.abs().max().item():.3f})")
        fx = {"meta_kind": np.array(kind), "meta_status": np.array(status), "config_json": np.array(json.dumps(cfg_json)),
              "input_ids": ids.numpy(), "n_masks": np.array(len(masks) if kind != "opt" else 0)}
        for i, m in enumerate(masks if kind != "opt" else []):
            fx[f"mask_{i}"] = m.numpy()
        for k, v in state.items():
            fx["w:" + k] = bits(v)
            fx["dtype:" + k] = np.array(str(v.dtype))
        if logits_ref is not None:
            fx["logits_reference"] = logits_ref.numpy()
        fx["logits_engine_torch_path"] = logits_mine.numpy()
        np.savez_compressed(os.path.join(a.out, f"ckpt_{kind}.npz"), **fx)


if __name__ == "__main__":
    main()
