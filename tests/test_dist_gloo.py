"""CPU, world_size 2 over gloo: the N > 1 path -- block partition of layers, one padded all-gather of the packed
records, artefacts of the other rank written into temp_storage_dir."""
import os
import socket
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layer(i):
    g = torch.Generator().manual_seed(100 + i)
    r = 4 + i  # per-layer ranks differ -> records of different length
    t = {"up": torch.randn(r, 8, generator=g).bfloat16(), "gate": torch.randn(r, 8, generator=g).bfloat16(),
         "down": torch.randn(8, r, generator=g).bfloat16(), "q_proj": torch.randn(2 * r, 8, generator=g).bfloat16(),
         "k_proj": torch.randn(r, 8, generator=g).bfloat16(), "v_proj": torch.randn(r, 8, generator=g).bfloat16(),
         "o_proj": torch.randn(8, 2 * r, generator=g).bfloat16()}
    return t, torch.arange(r).reshape(1, r) + i


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modegpt_amd import sharding as S
    from modegpt_amd.adapters.CompressionConfig import CompressionConfig

    class A:
        config = CompressionConfig(temp_storage_dir=os.path.join(tmp, f"rank{rank}"))

        def save_layer(self, output_dir, suffix, weights, layer_idx):
            os.makedirs(output_dir, exist_ok=True)
            torch.save(weights, os.path.join(output_dir, f"layer_{layer_idx}_{suffix}"))

    ad = A()
    chunk = list(range(5))  # 5 layers over 2 ranks: 3 + 2 (ragged)
    mine = S.my_layers(chunk, rank, world)
    assert mine == ([0, 1, 2] if rank == 0 else [3, 4])
    masks = []
    for i in mine:
        t, m = _layer(i)
        ad.save_layer(ad.config.temp_storage_dir, "mlp", {k: t[k] for k in ("up", "gate", "down")}, i)
        ad.save_layer(ad.config.temp_storage_dir, "qk", {k: t[k] for k in ("q_proj", "k_proj")}, i)
        ad.save_layer(ad.config.temp_storage_dir, "vo", {k: t[k] for k in ("v_proj", "o_proj")}, i)
        masks.append(m)
    all_masks = S.gather_layer_artifacts(ad, chunk, mine, masks, rank, world)
    assert len(all_masks) == 5
    for i in chunk:
        t, m = _layer(i)
        assert torch.equal(all_masks[i], m)
        got = {}
        for suffix in ("mlp", "qk", "vo"):
            got.update(torch.load(os.path.join(ad.config.temp_storage_dir, f"layer_{i}_{suffix}")))
        for k, v in t.items():
            assert torch.equal(got[k], v), (rank, i, k)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
