"""Module-level constants and HF model IO of the reference's src/model_utils.py.

The constants are read by the hot path exactly as in the reference (model_utils.py:15-31).  Model IO is HF
plumbing (network / disk bound, SURVEY.md section 2 "OUT OF SCOPE"): kept API-compatible, not accelerated.
"""
from __future__ import annotations

import logging
import os
import shutil

import torch

logger = logging.getLogger("MoDeGPT")

dtype_p = torch.float64   # precision of the statistics and factorizations   (model_utils.py:15)
dtype_f = torch.float16   # (unused legacy) final cast type                   (model_utils.py:19)
parallel = False          # reference's disabled two-GPU stub                 (model_utils.py:26)
conservative = True
d1 = "cuda:0"
d2 = "cuda:1" if parallel else "cuda:0"
calib_device = "cuda:1" if parallel else "cuda:0"

_REBUILD_FILE = {"opt": "OPTRebuild.py", "llama": "LlamaRebuild.py"}


def start_memory_usage_worker(path: str = "./.mem-usage", period_s: float = 1.0):
    """Daemon thread writing process RSS to ./.mem-usage once a second (model_utils.py:34-60)."""
    import threading
    import time

    import psutil

    def run():
        proc = psutil.Process(os.getpid())
        while True:
            rss = proc.memory_info().rss / 2 ** 30
            with open(path, "w") as f:
                f.write(f"[Monitor] Process RAM: {rss:.2f} GB\nSystem RAM: {psutil.virtual_memory().percent}% used")
            time.sleep(period_s)

    t = threading.Thread(target=run, daemon=True)
    t.start()
    return t


def _fix_pad(tokenizer):
    if tokenizer is not None and tokenizer.pad_token is None:
        tokenizer.pad_token = tokenizer.eos_token
        logger.info("No pad_token found. Set pad_token = eos_token.")


def load_model(model_name: str, device: int = 0):
    """model_utils.py:63-80."""
    from transformers import AutoModelForCausalLM, AutoTokenizer
    logger.info(f"Loading model from: {model_name}")
    tokenizer = AutoTokenizer.from_pretrained(model_name)
    model = AutoModelForCausalLM.from_pretrained(model_name, device_map="auto", trust_remote_code=True,
                                                 torch_dtype="auto")
    _fix_pad(tokenizer)
    return model, tokenizer, model.config


def rebuild_file_for(arch: str) -> str:
    if arch in _REBUILD_FILE:
        return _REBUILD_FILE[arch]
    if "qwen" in arch:
        return "DenseQwenRebuild.py"
    raise Exception("Cannot save compressed model ... no compressed model definition")


def save_compressed_model(adapter, rotary_masks, save_dir: str, source_model_name: str,
                          patchers_dir: str = "./src/patchers"):
    """Checkpoint writer with the reference's artefact set (model_utils.py:83-126): pytorch_model*.bin
    (safe_serialization=False), tokenizer, rotary_masks.pt, config.mask_path (absolute), torch_dtype=bfloat16, the
    arch's *Rebuild.py copied next to the weights, tokenizer_source.txt.  The *Rebuild.py modeling files are the
    reference's own (inference-time, out of scope here): they are copied from `patchers_dir` when present."""
    model, tokenizer = adapter.model, adapter.tokenizer
    rebuild_path = os.path.join(patchers_dir, rebuild_file_for(adapter.arch))
    os.makedirs(save_dir, exist_ok=True)
    if rotary_masks is not None:
        mask_path = os.path.abspath(os.path.join(save_dir, "rotary_masks.pt"))
        model.config.mask_path = mask_path
    else:
        mask_path = None
        model.config.mask_path = None
    model.config.torch_dtype = "bfloat16"
    model.config.dtype = "bfloat16"
    logger.info(f"Saving compressed model to {save_dir}")
    model.save_pretrained(save_dir, safe_serialization=False)
    if tokenizer is not None:
        tokenizer.save_pretrained(save_dir)
    if rotary_masks is not None:
        torch.save(rotary_masks, mask_path)
    if os.path.exists(rebuild_path):
        shutil.copy(rebuild_path, save_dir)
    else:
        logger.warning(f"{rebuild_path} not found: checkpoint written without its modeling file "
                       "(run from the reference checkout, or pass patchers_dir)")
    with open(os.path.join(save_dir, "tokenizer_source.txt"), "w") as f:
        f.write(source_model_name.strip())
    logger.info(f"Model, tokenizer, and tokenizer_source.txt saved to {save_dir}")


def reload_compressed_model(model_dir: str, device="cuda:0", tokenizer_source: str = ""):
    """model_utils.py:129-165."""
    from transformers import AutoModelForCausalLM, AutoTokenizer
    logger.info(f"Reloading compressed model from: {model_dir}")
    if not tokenizer_source:
        p = os.path.join(model_dir, "tokenizer_source.txt")
        tokenizer_source = open(p).read().strip() if os.path.exists(p) else model_dir
    tokenizer = AutoTokenizer.from_pretrained(tokenizer_source)
    model = AutoModelForCausalLM.from_pretrained(model_dir, trust_remote_code=True, device_map="auto",
                                                 torch_dtype="auto")
    _fix_pad(tokenizer)
    model.to(device)
    model.eval()
    return model, tokenizer
