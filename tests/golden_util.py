"""Load tests/golden/*.npz (written by oracle/gen_golden.py from the imported reference) as torch tensors."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_gqa", "tiny_gqa_k06", "tiny_mha", "tiny_opt", "med_gqa"]


def bf16_from_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.arch = str(z["meta_arch"])
        (self.d, self.d_ff, self.n_h, self.n_kv, self.hd, self.tokens, self.n_texts) = [int(v) for v in z["meta_dims"]]
        self.keep = float(z["meta_keep"])
        r = z["meta_ridges"]
        self.ridges = {"nystrom_ridge": float(r[0]), "ridge_qk": float(r[1]), "ridge_vo": float(r[2])}
        self.W = {k[2:]: bf16_from_bits(z[k]) for k in z.files if k.startswith("w_")}
        self.act = {k[4:]: bf16_from_bits(z[k]) for k in z.files if k.startswith("act_")}
        self.f64 = {k: torch.from_numpy(z[k]) for k in ("sigma_mlp", "sigma_x", "sigma_q", "sigma_k", "mlp_scores",
                                                         "mlp_down_f64", "sqrt_x", "invsqrt_x", "vo_v_f64", "vo_o_f64")}
        self.bf = {k: bf16_from_bits(z[k]) for k in ("mlp_up", "mlp_gate", "mlp_down", "qk_q", "qk_k", "vo_v", "vo_o")}
        self.mlp_idx = torch.from_numpy(z["mlp_idx"])
        self.mlp_rank = int(z["mlp_rank"])
        self.qk_rank = int(z["qk_rank"])
        self.qk_mask = torch.from_numpy(z["qk_mask"])
        self.vo_rank = int(z["vo_rank"])


def load_misc():
    return np.load(os.path.join(GOLDEN_DIR, "misc.npz"))


def vo_products(v, o, n_h, n_kv, r):
    """Sign/rotation-invariant view of the VO factors: per query head o'[:, head] @ v'[kv(head)]."""
    g = n_h // n_kv
    out = []
    for qh in range(n_h):
        h = qh // g
        out.append(o[:, qh * r:(qh + 1) * r].double() @ v[h * r:(h + 1) * r].double())
    return torch.stack(out)


def canon_rows(v, r):
    """Flip each row of a [n*r, d] factor so its largest-|.| entry is positive; returns (v', signs)."""
    i = v.abs().argmax(dim=1)
    s = torch.sign(v[torch.arange(v.shape[0]), i])
    s[s == 0] = 1
    return v * s[:, None], s


ROPE_CASES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "bf16_full": torch.bfloat16}


class RopeCase:
    """One entry of rope.npz: inputs and the reference's outputs of apply_rotary_pos_emb (masked) / _masked_rms_norm."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, "rope.npz"))
        self.dtype = ROPE_CASES[name]

        def t(key):
            a = z[f"{name}_{key}"]
            return torch.from_numpy(a.view(np.int16).copy()).view(self.dtype) if a.dtype == np.uint16 else torch.from_numpy(a)
        self.B, self.T, self.n_h, self.n_kv, self.hd, self.r = [int(v) for v in z[f"{name}_dims"]]
        self.q, self.k, self.cos, self.sin, self.q_out, self.k_out = (t(k) for k in ("q", "k", "cos", "sin", "q_out", "k_out"))
        m = z[f"{name}_mask"]
        self.mask = None if m.size == 0 else torch.from_numpy(m)
        self.has_norm = f"{name}_norm_w" in z.files
        if self.has_norm:
            self.norm_w, self.nq, self.nk = t("norm_w"), t("nq"), t("nk")


# ---------------------------------------------------------------- compressed checkpoints (oracle/gen_checkpoint_golden.py)
CKPT_KINDS = {"llama": "LlamaRebuild.py", "qwen3": "DenseQwenRebuild.py", "opt": "OPTRebuild.py"}


def materialise_checkpoint(kind: str, dst: str):
    """Rebuild the checkpoint directory of tests/golden/ckpt_<kind>.npz under `dst` exactly as save_compressed_model lays it
    out (config.json with an absolute mask_path, weights, rotary_masks.pt, the architecture's *Rebuild.py and the
    compressed_attention.py it imports).  Returns (directory, input_ids, fixture)."""
    import json
    import shutil
    from safetensors.torch import save_file
    z = np.load(os.path.join(GOLDEN_DIR, f"ckpt_{kind}.npz"))
    os.makedirs(dst, exist_ok=True)
    cfg = json.loads(str(z["config_json"]))
    masks = [torch.from_numpy(z[f"mask_{i}"]) for i in range(int(z["n_masks"]))]
    if masks:
        cfg["mask_path"] = os.path.abspath(os.path.join(dst, "rotary_masks.pt"))
        torch.save(masks, cfg["mask_path"])
    with open(os.path.join(dst, "config.json"), "w") as f:
        json.dump(cfg, f)
    state = {}
    for key in z.files:
        if key.startswith("w:"):
            a = z[key]
            state[key[2:]] = bf16_from_bits(a).clone() if str(z["dtype:" + key[2:]]) == "torch.bfloat16" else torch.from_numpy(a).clone()
    save_file(state, os.path.join(dst, "model.safetensors"), metadata={"format": "pt"})
    patchers = os.path.join(os.path.dirname(GOLDEN_DIR), os.pardir, "modegpt_amd", "patchers")
    for fname in (CKPT_KINDS[kind], "compressed_attention.py"):
        shutil.copy(os.path.join(patchers, fname), dst)
    return dst, torch.from_numpy(z["input_ids"]), z


LAYER_DRIVE = '''
def layer_drive(model, ids):
    """Logits by driving the model's own modules in order with an explicit additive causal mask (what the fixture's
    reference logits were produced with: oracle/gen_checkpoint_golden.py forward_through_layers)."""
    import torch
    core = model.model
    with torch.no_grad():
        h = core.embed_tokens(ids)
        B, T = ids.shape
        pos = torch.arange(T, device=ids.device)[None].expand(B, T)
        cos_sin = core.rotary_emb(h, pos)
        mask = torch.full((T, T), torch.finfo(h.dtype).min, dtype=h.dtype, device=ids.device).triu(1)[None, None]
        for layer in core.layers:
            out = layer(h, attention_mask=mask, position_ids=pos, position_embeddings=cos_sin)
            h = out[0] if isinstance(out, tuple) else out
        return model.lm_head(core.norm(h)).float()
'''
exec(LAYER_DRIVE)

OPT_DRIVE = '''
def opt_drive(model, ids):
    """OPT: the decoder's own modules in the order OPTDecoder.forward runs them, explicit additive causal mask, tied lm_head
    (what ckpt_opt.npz's reference logits were produced with: oracle/gen_checkpoint_golden.py opt_reference_logits)."""
    import torch
    dec = model.model.decoder
    with torch.no_grad():
        B, T = ids.shape
        am = torch.ones(B, T, dtype=torch.long, device=ids.device)
        pos = (torch.cumsum(am, dim=1) * am - 1).long()
        h = dec.embed_tokens(ids)
        if dec.project_in is not None:
            h = dec.project_in(h)
        h = h + dec.embed_positions(am, 0, position_ids=pos)
        mask = torch.full((T, T), torch.finfo(h.dtype).min, dtype=h.dtype, device=ids.device).triu(1)[None, None]
        for layer in dec.layers:
            out = layer(h, attention_mask=mask)
            h = out[0] if isinstance(out, tuple) else out
        if dec.final_layer_norm is not None:
            h = dec.final_layer_norm(h)
        if dec.project_out is not None:
            h = dec.project_out(h)
        return model.lm_head(h).float()
'''
exec(OPT_DRIVE)
