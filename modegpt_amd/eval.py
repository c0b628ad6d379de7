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


def load_c4(tokenizer, texts, n_samples):
    """n_samples random 2048-token windows, each from a randomly drawn document longer than 2048 tokens
    (eval.py:17-30; random.seed(1234) fixes the draw)."""
    random.seed(1234)
    seqlen = 2048
    windows = []
    while len(windows) < n_samples:
        while True:
            ids = tokenizer(texts[random.randint(0, len(texts) - 1)], return_tensors="pt").input_ids
            if ids.shape[1] > seqlen:
                break
        start = random.randint(0, ids.shape[1] - seqlen - 1)
        windows.append(ids[:, start:start + seqlen].to(device="cuda"))
    return windows


def _alpaca_prompt(example, with_output: bool) -> str:
    head = "Below is an instruction that describes a task"
    tail = example["output"] if with_output else ""
    if example.get("input", ""):
        return (f"{head}, paired with an input that provides further context. Write a response that appropriately "
                f"completes the request.\n\n### Instruction:\n{example['instruction']}\n\n### Input:\n{example['input']}"
                f"\n\n### Response:\n{tail}")
    return (f"{head}. Write a response that appropriately completes the request.\n\n### Instruction:\n"
            f"{example['instruction']}\n\n### Response:\n{tail}")


def get_alpaca_eval_data(n_samples: int = 500):
    """The last n_samples examples of tatsu-lab/alpaca's only split, formatted with their outputs (eval.py:228-255)."""
    from datasets import load_dataset
    ds = load_dataset("tatsu-lab/alpaca", split="train")
    ds = ds.select(range(len(ds) - n_samples, len(ds)))
    return [_alpaca_prompt(ex, with_output=True) for ex in ds]


def _eval_token_stream(tokenizer, dataset: str) -> torch.Tensor:
    """[1, n_tokens] ids of the evaluation text of `dataset`, tokenised in one call as upstream does (eval.py:141-160)."""
    from datasets import load_dataset
    if dataset == "wikitext":
        text = "\n\n".join(load_dataset("wikitext", "wikitext-2-raw-v1", split="test")["text"])
    elif dataset == "c4":
        raw = load_dataset("json", data_files={
            "validation": "https://huggingface.co/datasets/allenai/c4/resolve/main/en/c4-validation.00000-of-00008.json.gz"})
        text = "\n\n".join([t for t in raw["validation"]["text"] if len(t.strip()) > 0][:5000])
    elif dataset == "alpaca":
        text = "\n\n".join(get_alpaca_eval_data())
    else:
        raise ValueError(f"Unknown dataset: {dataset}. Must be 'wikitext', 'c4', or 'alpaca'")
    return tokenizer(text, return_tensors="pt").input_ids


@torch.no_grad()
def compute_perplexity(model, tokenizer, bs=16, device="cuda", dataset="wikitext", adapter=None):
    """Perplexity over consecutive 2048-token windows of the evaluation stream, at most 512 of them, `bs` per forward:
    exp(sum of token NLLs / (windows * 2047))  (eval.py:135-225).  Records throughput_tok/s and throughput_ktok/s in
    adapter.metrics.  Differences: `dataset="synthetic"` (seeded random ids, no network) and a window shorter than
    2048 for models whose max_position_embeddings is smaller (the tiny test models)."""
    model.eval()
    seqlen = _seq_len(model)
    if dataset == "synthetic":
        g = torch.Generator().manual_seed(4321)
        stream = torch.randint(0, model.config.vocab_size, (1, 8 * seqlen), generator=g)
    else:
        stream = _eval_token_stream(tokenizer, dataset)
    nsamples = min(stream.numel() // seqlen, 512)
    print(f"nsamples {nsamples}")
    nll_sum = torch.zeros((), dtype=torch.float32, device=device)
    tokens_done = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(0, nsamples, bs):
        j = min(i + bs, nsamples)
        inputs = stream[:, i * seqlen:j * seqlen].to(device).reshape(j - i, seqlen)
        logits = model(inputs).logits
        loss = torch.nn.functional.cross_entropy(logits[:, :-1, :].reshape(-1, logits.size(-1)), inputs[:, 1:].reshape(-1))
        nll_sum += loss.float() * (seqlen - 1) * (j - i)          # mean over the batch's predicted tokens, re-weighted
        tokens_done += (j - i) * seqlen
        if i:
            print(f"\rsample {i}/{nsamples} | ppl: {math.exp(nll_sum.item() / (i * (seqlen - 1))):.2f}", end="", flush=True)
    torch.cuda.synchronize()
    elapsed = max(time.perf_counter() - t0, 1e-9)
    print(f"\nThroughput: {tokens_done:,} tokens in {elapsed:.2f} s = {tokens_done / elapsed:,.0f} tok/s")
    if adapter:
        adapter.metrics["throughput_tok/s"] = tokens_done / elapsed
        adapter.metrics["throughput_ktok/s"] = tokens_done / elapsed / 1000
    ppl = math.exp(nll_sum.item() / max(nsamples * (seqlen - 1), 1))
    torch.cuda.empty_cache()
    return ppl


@torch.no_grad()
def evaluate_perplexity_alpaca(model, tokenizer, device="cuda"):
    """Per-example perplexity on the alpaca hold-out: every formatted example truncated to 2048 tokens, loss from the
    model's own label shift, token-weighted mean, non-finite losses skipped (eval.py:258-300)."""
    texts = get_alpaca_eval_data()
    model.eval()
    total, n_tok = 0.0, 0
    print(f"Evaluating Perplexity on {len(texts)} samples...")
    for text in texts:
        enc = tokenizer(text, return_tensors="pt", truncation=True, max_length=2048).to(device)
        loss = model(**enc, labels=enc["input_ids"].clone()).loss
        if not torch.isfinite(loss):
            print("Warning: Non-finite loss detected")
            continue
        n = enc["input_ids"].size(1)
        total += loss.item() * n
        n_tok += n
    return float("inf") if n_tok == 0 else math.exp(total / n_tok)
