"""Run-wide constants read by the hot path, and the HF checkpoint IO around it.

The constants keep the reference's names and values (src/model_utils.py:15-31) because adapters and compressors
import them by name.  The IO functions are API-compatible plumbing (HF load / save, network or disk bound --
SURVEY.md section 2 "OUT OF SCOPE"): nothing here is accelerated, it only has to write a checkpoint that loads through
config.auto_map -- with this engine's modeling files (patchers/*Rebuild.py, shipped by default) or the reference's.
"""
from __future__ import annotations

import logging
import os
import shutil
import threading
import time

import torch

logger = logging.getLogger("MoDeGPT")

# ---- precision / placement constants -------------------------------------------------------------------------
dtype_p = torch.float64      # statistics and factorizations
dtype_f = torch.float16      # legacy "final" dtype (the artefacts are bf16)
parallel = False             # the reference's two-GPU stub; multi-GPU here is sharding.py, not this flag
conservative = True


def local_rank() -> int:
    """This process's GPU on the node: LOCAL_RANK under torchrun / bench.py's own launcher, else 0."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def local_device() -> str:
    """Device string of this rank's GPU, evaluated at CALL time (one process per GPU: rank g works on cuda:g; the
    reference is single-process and says "cuda:0" everywhere, src/model_utils.py:25-31)."""
    return f"cuda:{local_rank()}"


# The reference's names, kept because adapters / compressors import them by name.  Under torchrun LOCAL_RANK is in the
# environment before this module is imported, so the constants already point at the rank's own GPU; code in this package
# that can run before the environment is final (tests spawning ranks in-process) calls local_device() instead.
d1 = local_device()
d2 = "cuda:1" if parallel else local_device()
calib_device = d2

# architecture -> modeling file the checkpoint ships with (this engine's own, modegpt_amd/patchers/; file and class
# names are the reference's so config.auto_map is unchanged)
PATCHERS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "patchers")
REBUILD_FILES = {"opt": "OPTRebuild.py", "llama": "LlamaRebuild.py", "qwen": "DenseQwenRebuild.py"}


def rebuild_file_for(arch: str) -> str:
    for key, fname in REBUILD_FILES.items():
        if key in arch:
            return fname
    raise Exception("Cannot save compressed model ... no compressed model definition")


def start_memory_usage_worker(path: str = "./.mem-usage", period_s: float = 1.0) -> threading.Thread:
    """1 Hz RSS monitor writing ./.mem-usage (daemon thread), as the reference's run does."""
    import psutil
    me = psutil.Process(os.getpid())

    def loop():
        while True:
            gib = me.memory_info().rss / 2 ** 30
            text = f"[Monitor] Process RAM: {gib:.2f} GB\nSystem RAM: {psutil.virtual_memory().percent}% used"
            if gib > 60:
                text += "\n\nCRITICAL WARNING: Process nearing 64GB RAM limit! Crash imminent.\n"
            with open(path, "w") as f:
                f.write(text)
            time.sleep(period_s)

    t = threading.Thread(target=loop, daemon=True)
    t.start()
    return t


def _causal_lm(source: str):
    """device_map="auto" as upstream when this is the only process; one process per GPU keeps the whole model on its own card."""
    from transformers import AutoModelForCausalLM
    sharded = int(os.environ.get("WORLD_SIZE", "1")) > 1
    device_map = {"": local_rank()} if (sharded and torch.cuda.is_available()) else "auto"
    return AutoModelForCausalLM.from_pretrained(source, device_map=device_map, trust_remote_code=True, torch_dtype="auto")


def _tokenizer(source: str):
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(source)
    if tok.pad_token is None:
        tok.pad_token = tok.eos_token
        logger.info("No pad_token found. Set pad_token = eos_token.")
    return tok


def load_model(model_name: str, device: int = 0):
    """(model, tokenizer, config) of an official checkpoint."""
    logger.info(f"Loading model from: {model_name}")
    model = _causal_lm(model_name)
    return model, _tokenizer(model_name), model.config


def reload_compressed_model(model_dir: str, device=None, tokenizer_source: str = ""):
    """Load an original OR compressed checkpoint; a compressed one names its tokenizer in tokenizer_source.txt and
    its modeling file through config.auto_map (trust_remote_code)."""
    logger.info(f"Reloading compressed model from: {model_dir}")
    marker = os.path.join(model_dir, "tokenizer_source.txt")
    if not tokenizer_source:
        tokenizer_source = open(marker).read().strip() if os.path.exists(marker) else model_dir
    model = _causal_lm(model_dir).to(device or local_device()).eval()
    return model, _tokenizer(tokenizer_source)


def save_compressed_model(adapter, rotary_masks, save_dir: str, source_model_name: str,
                          patchers_dir: str = PATCHERS_DIR):
    """Write the artefact set the reference's loader expects: pytorch_model*.bin (safe_serialization=False is passed as
    upstream does; transformers >= 5 ignores it and writes model.safetensors, which the same loader reads),
    tokenizer files, rotary_masks.pt + config.mask_path (absolute), config dtype bfloat16, the architecture's
    *Rebuild.py next to the weights, tokenizer_source.txt."""
    model, tokenizer = adapter.model, adapter.tokenizer
    os.makedirs(save_dir, exist_ok=True)
    mask_path = os.path.abspath(os.path.join(save_dir, "rotary_masks.pt")) if rotary_masks is not None else None
    model.config.mask_path = mask_path
    model.config.torch_dtype = model.config.dtype = "bfloat16"
    logger.info(f"Saving compressed model to {save_dir}")
    model.save_pretrained(save_dir, safe_serialization=False)
    if tokenizer is not None:
        tokenizer.save_pretrained(save_dir)
    if mask_path is not None:
        torch.save(rotary_masks, mask_path)
    # the modeling file config.auto_map names, and the module it imports relatively: together they depend on torch and
    # transformers only, so the checkpoint loads on a machine without this engine (as the reference's single file does)
    shutil.copy(os.path.join(patchers_dir, rebuild_file_for(adapter.arch)), save_dir)
    support = os.path.join(patchers_dir, "compressed_attention.py")
    if os.path.exists(support):
        shutil.copy(support, save_dir)
    with open(os.path.join(save_dir, "tokenizer_source.txt"), "w") as f:
        f.write(source_model_name.strip())
    logger.info(f"Model, tokenizer, and tokenizer_source.txt saved to {save_dir}")
