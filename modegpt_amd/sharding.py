"""Layer sharding across the GPUs of one node (new: the reference is single-process).

Given the sigma matrices, every layer's MLP / QK / VO compression is independent, so rank g of G owns a
contiguous block of layers (equal blocks, or blocks balanced against the forward a rank has to run to reach them:
partition()) and no data-path collective is needed until the end, when ONE all-gather
(RCCL over xGMI under backend "nccl", gloo on CPU in tests) of a fixed-stride packed buffer reassembles the
compressed checkpoint on every rank.

Packed record per layer (all 2-byte words, bf16 bit patterns or int16 quarters of int64s):
    header  : 16 x int64  = [layer_idx, n_tensors, (rows, cols) x 7 tensors]   (up, gate, down, q, k, v, o)
    mask hdr:  2 x int64  = [n_kv, rank]  followed by the int64 rotary mask
    payload : the tensors' bf16 words back to back
Records are padded to the largest record of the call (per-layer ranks differ with the keep ratios); a rank sends as many
records as it owns layers (the counts travel with the stride agreement), so uneven blocks do not pad to the largest block.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

TENSOR_ORDER = ("up", "gate", "down", "q_proj", "k_proj", "v_proj", "o_proj")
_HDR_I64 = 16


def init_from_env() -> Tuple[int, int]:
    """(rank, world).  Initialises torch.distributed when launched by torchrun; backend nccl (= RCCL) on GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def finalize() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def forward_share() -> float:
    """Cost of taking the calibration samples through ONE layer of the model forward, relative to that layer's covariance +
    decomposition + artefact IO (env MODEGPT_SHARD_FORWARD_SHARE, default 0: equal blocks).  run_modegpt sets it to its measured
    0.32 (random-init Llama-3-8B, 512 x 2048 tokens on one MI355X: forward 15.4 s against 48.2 s for 32 layers)."""
    return float(os.environ.get("MODEGPT_SHARD_FORWARD_SHARE", "0"))


def partition(n: int, world: int, share: Optional[float] = None) -> List[Tuple[int, int]]:
    """Contiguous blocks [start, end) of n layers for ranks 0 .. world - 1.
    share == 0: equal blocks of ceil(n / world) (the synthetic bench: no forward).  share > 0: a rank that owns layers [a, b) pays
    the forward up to b (it stops there) plus its own layers' covariance / decomposition, f b + c (b - a) with f / c = share;
    equalising that over the ranks gives b_g = n (1 - r^g) / (1 - r^world), r = 1 / (1 + share): early ranks own more layers, the
    last ones -- which traverse (almost) the whole model -- few."""
    share = forward_share() if share is None else share
    if world <= 1:
        return [(0, n)]
    if share <= 0:
        per = -(-n // world)
        return [(min(n, g * per), min(n, (g + 1) * per)) for g in range(world)]
    r = 1.0 / (1.0 + share)
    bounds = [0] + [int(round(n * (1 - r ** g) / (1 - r ** world))) for g in range(1, world)] + [n]
    for g in range(1, world + 1):                      # monotone, and nobody beyond n
        bounds[g] = min(n, max(bounds[g], bounds[g - 1]))
    return [(bounds[g], bounds[g + 1]) for g in range(world)]


def my_layers(layers: Sequence[int], rank: int, world: int, share: Optional[float] = None) -> List[int]:
    """This rank's contiguous block of `layers` (partition())."""
    layers = list(layers)
    a, b = partition(len(layers), world, share)[rank]
    return layers[a:b]


def max_block(n: int, world: int, share: Optional[float] = None) -> int:
    """Records per rank in the all-gather: the largest block of the partition."""
    return max(b - a for a, b in partition(n, world, share))


def owner_of(pos: int, n: int, world: int, share: Optional[float] = None) -> int:
    for g, (a, b) in enumerate(partition(n, world, share)):
        if a <= pos < b:
            return g
    raise IndexError(pos)


# ------------------------------------------------------------------ pack / unpack
def pack_layer(layer_idx: int, tensors: Dict[str, Optional[torch.Tensor]], mask: Optional[torch.Tensor]) -> torch.Tensor:
    """One layer's compressed tensors (+ rotary mask) -> flat int16 record on the tensors' device."""
    dev = next(t.device for t in tensors.values() if t is not None)
    hdr = torch.zeros(_HDR_I64, dtype=torch.int64)
    hdr[0] = layer_idx
    parts = []
    n = 0
    for slot, name in enumerate(TENSOR_ORDER):
        t = tensors.get(name)
        if t is None:
            continue
        if t.dtype != torch.bfloat16 or t.dim() != 2:
            raise ValueError(f"{name}: expected a 2-D bf16 tensor")
        hdr[2 + 2 * slot], hdr[3 + 2 * slot] = t.shape[0], t.shape[1]
        parts.append(t.contiguous().view(torch.int16).reshape(-1))
        n += 1
    hdr[1] = n
    if mask is None:
        mh = torch.zeros(2, dtype=torch.int64)
        mparts = []
    else:
        mh = torch.tensor(list(mask.shape), dtype=torch.int64)
        mparts = [mask.to(torch.int64).contiguous().view(torch.int16).reshape(-1)]
    head = torch.cat([hdr, mh]).view(torch.int16).to(dev)
    return torch.cat([head] + [p.to(dev) for p in mparts] + parts)


def unpack_layer(rec: torch.Tensor) -> Tuple[int, Dict[str, torch.Tensor], Optional[torch.Tensor]]:
    words_hdr = (_HDR_I64 + 2) * 4
    head = rec[:words_hdr].cpu().view(torch.int64)
    layer_idx, off = int(head[0]), words_hdr
    n_kv, r = int(head[_HDR_I64]), int(head[_HDR_I64 + 1])
    mask = None
    if n_kv * r > 0:
        mask = rec[off:off + n_kv * r * 4].contiguous().view(torch.int64).reshape(n_kv, r)
        off += n_kv * r * 4
    out = {}
    for slot, name in enumerate(TENSOR_ORDER):
        rows, cols = int(head[2 + 2 * slot]), int(head[3 + 2 * slot])
        if rows * cols == 0:
            continue
        out[name] = rec[off:off + rows * cols].contiguous().view(torch.bfloat16).reshape(rows, cols)
        off += rows * cols
    return layer_idx, out, mask


def gather_buffers(per_rank: int, world: int, record_words: int, device) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Send / receive buffers for allgather_records, allocated ahead of time (bench.py: outside its timed region -- the receive
    side is world x per_rank x 0.34 GB at Llama-3-8B shapes).  record_words: an upper bound of a record's length in int16 words."""
    words = per_rank * (4 + (record_words + 3) // 4 * 4)
    send = torch.empty(words, dtype=torch.int16, device=device)
    recv = torch.empty(world * words, dtype=torch.int16, device=device) if world > 1 else None
    return send, recv


def _allgather_rows(send: torch.Tensor, recv: torch.Tensor, counts: List[int], rank: int) -> None:
    """recv = the ranks' row blocks back to back, rank g contributing counts[g] rows of send's width.  ONE collective either way:
    equal counts -> all_gather_into_tensor (ncclAllGather; what bench.py's equal blocks take); ragged counts -> torch's all_gather
    with unevenly sized outputs, which the nccl backend runs as one coalesced group of broadcasts (an all-gather-v: every rank's
    block crosses each xGMI link once, nothing is padded to the largest block).  gloo has no uneven all-gather: the CPU tests walk
    the ranks with one broadcast each -- same offsets, same result, test plumbing only."""
    width = send.shape[1]
    if len(set(counts)) == 1:
        dist.all_gather_into_tensor(recv.view(torch.uint8), send.view(torch.uint8))     # byte views: gloo has no int16
        return
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    outs = [recv[offs[g]:offs[g + 1]].view(torch.uint8).reshape(-1) for g in range(len(counts))]
    mine = send.view(torch.uint8).reshape(-1)
    assert mine.numel() == counts[rank] * width * 2
    if dist.get_backend() == "nccl":
        dist.all_gather(outs, mine)
    else:
        for g in range(len(counts)):
            if g == rank:
                outs[g].copy_(mine)
            dist.broadcast(outs[g], src=g)


def allgather_records(records: List[torch.Tensor], per_rank: int, world: int, force_collective: bool = False,
                      buffers: Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]] = None) -> List[torch.Tensor]:
    """The single data-path collective: every rank contributes ITS records (at most `per_rank`), each padded to the common record
    stride; a rank with fewer records than another sends fewer rows -- the ranks' counts travel in the control message that
    already agrees on the stride (one MAX all-reduce of 8 (world + 1) bytes), so the balanced partition of a real model
    (9, 6, 5, 4, 3, 2, 2, 1 layers at 8 ranks) moves 32 records, not 8 x 9.
    force_collective runs the all-gather even at world == 1 (lets a 1-GPU box exercise the RCCL code path).
    buffers: gather_buffers() of the caller's, used when they are large enough for the stride the ranks agree on."""
    dev = records[0].device if records else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    collective = world > 1 or force_collective
    rank = dist.get_rank() if (collective and dist.is_initialized()) else 0
    if len(records) > per_rank:
        raise ValueError(f"{len(records)} records for {per_rank} slots")
    ctrl = torch.zeros(world + 1, dtype=torch.int64, device=dev)
    ctrl[0] = max([r.numel() for r in records] + [0])
    ctrl[1 + rank] = len(records)
    if collective:
        dist.all_reduce(ctrl, op=dist.ReduceOp.MAX)    # size agreement (stride + every rank's record count), not a data-path exchange
    ctrl = ctrl.cpu().tolist()
    stride = (int(ctrl[0]) + 3) // 4 * 4                 # rows stay 8-byte aligned for the int64 header views
    counts = [int(c) for c in ctrl[1:]]
    if len(set(counts)) > 1:
        counts = [max(c, 1) for c in counts]             # (ragged: an empty rank still sends one unused row -- no zero-size operands)
    PRE = 4                                                 # words 0..3: slot-in-use flag (+ alignment pad)
    rows = counts[rank]
    words = rows * (PRE + stride)
    own_send, own_recv = buffers if buffers is not None else (None, None)
    if own_send is not None and own_send.numel() >= words and own_send.device == dev:
        send = own_send[:words].view(rows, PRE + stride)
    else:
        send = torch.empty(rows, PRE + stride, dtype=torch.int16, device=dev)
    send[:, :PRE] = 0
    for i, r in enumerate(records):
        send[i, 0] = 1
        send[i, PRE:PRE + r.numel()] = r
        send[i, PRE + r.numel():] = 0                      # (padding up to the common stride: defined bytes on the wire)
    if not collective:
        return [send[i, PRE:] for i in range(rows) if send[i, 0] == 1]
    total = sum(counts)
    if own_recv is not None and own_recv.numel() >= total * (PRE + stride) and own_recv.device == dev:
        recv = own_recv[:total * (PRE + stride)].view(total, PRE + stride)
    else:
        recv = torch.empty(total, PRE + stride, dtype=torch.int16, device=dev)
    _allgather_rows(send, recv, counts, rank)
    used = recv[:, 0].cpu()
    return [recv[i, PRE:] for i in range(total) if used[i] == 1]


def gather_layer_artifacts(adapter, chunk: Sequence[int], mine: Sequence[int], rotary_masks, rank: int, world: int,
                           writer_rank: int = 0):
    """After this rank compressed `mine`: all-gather everybody's layers of `chunk` and return the chunk's rotary masks in
    layer order.  ONLY `writer_rank` -- the rank that goes on to convert_model (run_modegpt.py) -- writes the artefacts it did
    not produce into temp_storage_dir, so that it finds layer_<i>_{mlp,qk,vo} for every i.  All ranks may share one
    temp_storage_dir (they do under torchrun: same --temp_storage_dir flag): every file has exactly one writer at any time --
    its owner before the all-gather (which orders the owner's write before any other rank's access), then at most
    writer_rank, through a temporary name + os.replace so that a reader never sees a half-written file."""
    rms = list(rotary_masks or [])
    if world == 1:
        return rms
    # The send records are packed from the tensors save_layer was handed -- still on the device when the run asked the adapter to
    # hold them (adapter.hold_artifacts(True): run_modegpt does for sharded runs) -- so the compressed layers go HBM -> xGMI, not
    # HBM -> host -> disk -> host -> HBM.  An adapter that holds nothing (a duck-typed one; a resumed run) reads its files back.
    held = getattr(adapter, "take_held_artifacts", None)
    flush = getattr(adapter, "flush_artifacts", None)
    if flush is not None:
        flush()         # a background artefact writer: this rank's own files are complete before anybody else can touch their names
    d = os.path.expandvars(adapter.config.temp_storage_dir)
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    records = []
    for pos, layer in enumerate(mine):
        tensors = held(layer) if held is not None else {}
        if not tensors:
            for suffix in ("mlp", "qk", "vo"):
                p = os.path.join(d, f"layer_{layer}_{suffix}")
                if os.path.exists(p):
                    tensors.update(torch.load(p, map_location=dev))
        records.append(pack_layer(layer, tensors, rms[pos] if pos < len(rms) else None))
    per = max_block(len(chunk), world)
    masks = {}
    for rec in allgather_records(records, per, world):
        layer, tensors, mask = unpack_layer(rec)
        masks[layer] = mask
        if layer in mine or rank != writer_rank:
            continue
        groups = {"mlp": ("up", "gate", "down"), "qk": ("q_proj", "k_proj"), "vo": ("v_proj", "o_proj")}
        for suffix, names in groups.items():
            w = {k: tensors[k].clone() for k in names if k in tensors}
            if w:
                _write_artifact(d, layer, suffix, w)
    return [masks[l] for l in chunk if masks.get(l) is not None]


def _write_artifact(directory: str, layer: int, suffix: str, weights: Dict[str, torch.Tensor]) -> None:
    """save_layer's file (model_adapter.py:184-191: torch.save of {name: bf16 tensor} as layer_<i>_<suffix>), written
    atomically."""
    os.makedirs(directory, exist_ok=True)
    final = os.path.join(directory, f"layer_{layer}_{suffix}")
    tmp = f"{final}.tmp{os.getpid()}"
    torch.save(weights, tmp)
    os.replace(tmp, final)
