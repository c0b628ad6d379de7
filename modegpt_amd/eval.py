"""Calibration-text loaders and perplexity (reference: src/eval.py).  HF datasets/network plumbing, not
accelerated (SURVEY.md section 2 "OUT OF SCOPE"); only the output contract matters to the hot path: a list of
[B, T<=2048] int64 batches on the GPU.  `dataset="synthetic"` (not upstream) draws seeded random token ids so the
path can run without network access."""
from __future__ import annotations

import logging
import math
import random
import time

import numpy as np
import torch

logger = logging.getLogger("MoDeGPT")


def _seq_len(model) -> int:
    return min(2048, getattr(model.config, "max_position_embeddings", 2048))


def chunk_text(model, tokenizer, long_texts: str, min_threshold):
    """eval.py:122-131."""
    ids = tokenizer(long_texts, truncation=False, return_tensors="pt", add_special_tokens=False)["input_ids"][0]
    L = _seq_len(model)
    n = ids.size(0) // L
    return ids[: n * L].view(-1, L)


def _batches(chunks: torch.Tensor, batch_size: int, device="cuda"):
    return [chunks[i:i + batch_size].to(device) for i in range(0, chunks.shape[0], batch_size)]


def load_synthetic_texts(calib_size, model, batch_size, seed=1234, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, model.config.vocab_size, (int(calib_size), _seq_len(model)), generator=g)
    return _batches(ids, batch_size, device)


def load_alpaca_texts(calib_size, model, tokenizer, batch_size, seed=1234):
    """eval.py:72-119."""
    from datasets import load_dataset

    def fmt(s):
        head = "Below is an instruction that describes a task"
        if s.get("input"):
            return (f"{head}, paired with an input that provides further context. Write a response that appropriately "
                    f"completes the request.\n\n### Instruction:\n{s['instruction']}\n\n### Input:\n{s['input']}\n\n"
                    f"### Response:\n") + tokenizer.eos_token
        return (f"{head}. Write a response that appropriately completes the request.\n\n### Instruction:\n"
                f"{s['instruction']}\n\n### Response:\n") + tokenizer.eos_token

    ds = load_dataset("tatsu-lab/alpaca", split="train").shuffle(seed=seed)
    L = _seq_len(model)
    need = L * calib_size
    toks = []
    for s in ds:
        toks.extend(tokenizer(fmt(s), return_tensors=None, add_special_tokens=False)["input_ids"])
        if len(toks) >= need:
            break
    ids = torch.tensor(toks[:need], dtype=torch.long).view(calib_size, L)
    return _batches(ids, batch_size)


def load_calibration_texts(calib_size, model, tokenizer, batch_size: int, dataset="wikitext"):
    """eval.py:33-69: concatenate the corpus, cut into max_length chunks, sample calib_size chunks without
    replacement under np.random.seed(1234), batch."""
    if dataset == "synthetic":
        return load_synthetic_texts(calib_size, model, batch_size)
    if dataset == "alpaca":
        return load_alpaca_texts(calib_size=calib_size, model=model, tokenizer=tokenizer, batch_size=batch_size)
    from datasets import load_dataset
    if dataset == "wikitext":
        text = "\n\n".join(load_dataset("wikitext", "wikitext-2-raw-v1", split="train")["text"])
    elif dataset == "c4":
        ds = load_dataset("json", data_files={
            "train": "https://huggingface.co/datasets/allenai/c4/resolve/main/en/c4-train.00000-of-01024.json.gz"})
        text = "\n\n".join([t for t in ds["train"]["text"] if len(t.strip()) > 0][:10000])
    else:
        raise ValueError(f"Unknown dataset: {dataset}. Must be 'wikitext' or 'c4'")
    chunks = chunk_text(model=model, tokenizer=tokenizer, long_texts=text, min_threshold=2048)
    np.random.seed(1234)
    pick = np.random.choice(chunks.shape[0], size=min(int(calib_size), chunks.shape[0]), replace=False)
    return _batches(chunks[pick], batch_size)


@torch.no_grad()
def compute_perplexity(model, tokenizer, dataset="wikitext", adapter=None, batch_size: int = 4, device="cuda"):
    """Token-level perplexity over the dataset's test split in max_length windows (eval.py:135-225); records
    throughput_tok/s in adapter.metrics."""
    L = _seq_len(model)
    if dataset == "synthetic":
        g = torch.Generator().manual_seed(4321)
        ids = torch.randint(0, model.config.vocab_size, (8, L), generator=g)
    else:
        from datasets import load_dataset
        if dataset == "c4":
            raw = load_dataset("json", data_files={
                "validation": "https://huggingface.co/datasets/allenai/c4/resolve/main/en/c4-validation.00000-of-00008.json.gz"})
            text = "\n\n".join(raw["validation"]["text"][:1100])
        else:
            text = "\n\n".join(load_dataset("wikitext", "wikitext-2-raw-v1", split="test")["text"])
        ids = chunk_text(model, tokenizer, text, min_threshold=L)
    model.eval()
    nll, count, t0 = 0.0, 0, time.time()
    for i in range(0, ids.shape[0], batch_size):
        b = ids[i:i + batch_size].to(device)
        logits = model(b).logits[:, :-1].float()
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), b[:, 1:].reshape(-1),
                                                 reduction="sum")
        nll += loss.item()
        count += b[:, 1:].numel()
    ppl = math.exp(nll / max(count, 1))
    if adapter is not None:
        adapter.metrics["throughput_tok/s"] = ids.numel() / max(time.time() - t0, 1e-9)
    return ppl
