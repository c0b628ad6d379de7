"""CPU, world_size 2 over gloo: the N > 1 path -- block partition of layers, one padded all-gather of the packed records, and
the artefact hand-over to rank 0 through ONE temp_storage_dir shared by all ranks (what torchrun gives them: the same
--temp_storage_dir flag), driven through run_modegpt.compress_chunk itself with the GPU stages replaced by writers of
deterministic artefacts.  Plus bench.py's own launcher (`python bench.py --gpus 2` without torchrun)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


N_LAYERS = 15     # three chunks of 5 layers; each chunk over 2 ranks: 3 + 2 (ragged)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modegpt_amd import model_utils, run_modegpt as R, sharding as S
    from modegpt_amd.adapters.CompressionConfig import CompressionConfig

    # one process per GPU: everything this rank allocates goes to ITS card, whatever the reference's "cuda:0" literals say
    assert model_utils.local_rank() == rank and model_utils.local_device() == f"cuda:{rank}"

    shared = os.path.join(tmp, "layers")          # ONE directory for all ranks

    class A:
        config = CompressionConfig(temp_storage_dir=shared, calib_size=4, calibs_batch_size=2, compression_ratio=0.3)
        n_layers = N_LAYERS

        def save_layer(self, output_dir, suffix, weights, layer_idx):
            os.makedirs(output_dir, exist_ok=True)
            torch.save(weights, os.path.join(output_dir, f"layer_{layer_idx}_{suffix}"))

    ad = A()
    calibrated = []

    # the GPU stages of compress_chunk, replaced: same signatures, deterministic artefacts written through adapter.save_layer
    def load_calibs(adapter, n_samples, batch_size, dataset, target_layers):
        calibrated.append(list(target_layers))
        stops.append((adapter.calib_stop_after, adapter.calib_want_bi))
        none = [None] * N_LAYERS
        # BI for ALL layers, but only from the rank that runs the whole forward (compress_chunk hands them to the others)
        return none, none, none, none, ([0.1 * (i + 1) for i in range(N_LAYERS)] if adapter.calib_want_bi else None)

    stops = []
    seen_bi = []

    def allocate(bi, **kw):
        seen_bi.append(list(bi))
        return [0.7] * N_LAYERS

    def nystrom(adapter, cov, keep_ratios, target_layers):
        for i in target_layers:
            adapter.save_layer(adapter.config.temp_storage_dir, "mlp", {k: _layer(i)[0][k] for k in ("up", "gate", "down")}, i)

    def qk(adapter, cov, keep_ratios, target_layers):
        for i in target_layers:
            adapter.save_layer(adapter.config.temp_storage_dir, "qk", {k: _layer(i)[0][k] for k in ("q_proj", "k_proj")}, i)
        return [_layer(i)[1] for i in target_layers]

    def vo(adapter, cov, keep_ratios, target_layers):
        for i in target_layers:
            adapter.save_layer(adapter.config.temp_storage_dir, "vo", {k: _layer(i)[0][k] for k in ("v_proj", "o_proj")}, i)

    R.load_calibs, R.allocate_global_sparsity, R.compress_nystrom, R.compress_qk, R.compress_vo = load_calibs, allocate, nystrom, qk, vo
    R._free = lambda: None
    written = []
    real_write = S._write_artifact
    S._write_artifact = lambda d, layer, suffix, w: (written.append((layer, suffix)), real_write(d, layer, suffix, w))[1]

    foreign = []
    for first in range(0, N_LAYERS, 5):               # run_modegpt's loop over chunks (LAYERS_PER_STEP there), back to back
        chunk = list(range(first, first + 5))
        mine = S.my_layers(chunk, rank, world)
        assert mine == (chunk[:3] if rank == 0 else chunk[3:])
        all_masks = R.compress_chunk(ad, ad.config, chunk, rank, world)
        assert calibrated[-1] == mine                 # hooks only for this rank's layers
        # the forward stops behind this rank's last layer; the last rank computes the BI scores, once (later chunks reuse them)
        assert stops[-1] == (mine[-1], rank == world - 1 and first == 0)
        assert all(abs(a - 0.1 * (i + 1)) < 1e-12 for i, a in enumerate(seen_bi[-1])) and len(seen_bi[-1]) == N_LAYERS
        assert len(all_masks) == 5
        for pos, i in enumerate(chunk):
            assert torch.equal(all_masks[pos], _layer(i)[1])
        foreign += [i for i in chunk if i not in mine]
        if rank == 0:
            # what convert_model does with the artefacts, with NO barrier in between (rank 1 is already in its next chunk):
            # every layer's three files must be whole and right
            for i in chunk:
                got = {}
                for suffix in ("mlp", "qk", "vo"):
                    got.update(torch.load(os.path.join(shared, f"layer_{i}_{suffix}")))
                for k, v in _layer(i)[0].items():
                    assert torch.equal(got[k], v), (i, k)
    # only the consumer wrote artefacts it does not own; the others wrote nothing beyond their own layers
    if rank == 0:
        assert sorted(written) == sorted((i, s) for i in foreign for s in ("mlp", "qk", "vo"))
    else:
        assert written == []
    # the gather into buffers of the caller's (bench.py allocates them before its timed region): same records, the receive side
    # inside the caller's buffer, ragged record lengths and a rank with fewer records than slots
    mine = [S.pack_layer(10 * rank + i, _layer(3 * rank + i)[0], _layer(3 * rank + i)[1]) for i in range(2 if rank == 0 else 1)]
    bufs = S.gather_buffers(2, world, 2000, "cpu")
    bufs[0].fill_(-1)
    bufs[1].fill_(-1)
    out = S.allgather_records(mine, 2, world, buffers=bufs)
    assert len(out) == 3 and out[0].untyped_storage().data_ptr() == bufs[1].untyped_storage().data_ptr()
    got = {S.unpack_layer(r)[0]: S.unpack_layer(r) for r in out}
    assert sorted(got) == [0, 1, 10]
    for li, src in ((0, 0), (1, 1), (10, 3)):
        _, tensors, mask = got[li]
        assert torch.equal(mask, _layer(src)[1]) and all(torch.equal(tensors[k], v) for k, v in _layer(src)[0].items())
    S.finalize()                                      # barrier + destroy: the collective phase ends here for every rank
    assert not dist.is_initialized()
    assert not [f for f in os.listdir(shared) if ".tmp" in f]     # (after the barrier: rank 0 has finished its atomic renames)


def test_two_rank_compress_chunk_with_a_shared_artefact_directory():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)


def _worker_balanced(rank, world, port, tmp):
    """Three ranks, ONE chunk of 5 layers under the forward-balanced partition (share 2 -> blocks of 3, 2 and 0 layers): the last
    rank owns nothing and only computes the BI scores; the send records come from the tensors ModelAdapter.save_layer HOLDS (no file
    is read back: torch.load is forbidden during the gather); the ranks send 3 / 2 / 1(unused) rows, not 3 x 3."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MODEGPT_SHARD_FORWARD_SHARE="2.0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modegpt_amd import engine, run_modegpt as R, sharding as S
    from modegpt_amd.adapters.CompressionConfig import CompressionConfig
    from modegpt_amd.adapters.model_adapter import ModelAdapter
    n = 5
    assert S.partition(n, world) == [(0, 3), (3, 5), (5, 5)]
    shared = os.path.join(tmp, "layers")

    class B(engine.TensorAdapter):                 # a concrete adapter with the REAL save_layer (files + held tensors)
        save_layer = ModelAdapter.save_layer

    ad = B(dict(engine.SHAPES["tiny"], n_layers=n), {})
    ad.config = CompressionConfig(temp_storage_dir=shared, calib_size=4, calibs_batch_size=2, compression_ratio=0.3)
    ad.async_artifacts(True)
    ad.hold_artifacts(True)
    seen = {}

    def load_calibs(adapter, n_samples, batch_size, dataset, target_layers):
        seen["calib"] = (list(target_layers), adapter.calib_stop_after, adapter.calib_want_bi, adapter.calib_no_hooks)
        none = [None] * n
        return none, none, none, none, ([0.5 + i for i in range(n)] if adapter.calib_want_bi else None)

    def allocate(bi, **kw):
        seen["bi"] = list(bi)
        return [0.7] * n

    def nystrom(adapter, cov, keep_ratios, target_layers):
        for i in target_layers:
            adapter.save_layer(adapter.config.temp_storage_dir, "mlp", {k: _layer(i)[0][k] for k in ("up", "gate", "down")}, i)

    def qk(adapter, cov, keep_ratios, target_layers):
        for i in target_layers:
            adapter.save_layer(adapter.config.temp_storage_dir, "qk", {k: _layer(i)[0][k] for k in ("q_proj", "k_proj")}, i)
        return [_layer(i)[1] for i in target_layers]

    def vo(adapter, cov, keep_ratios, target_layers):
        for i in target_layers:
            adapter.save_layer(adapter.config.temp_storage_dir, "vo", {k: _layer(i)[0][k] for k in ("v_proj", "o_proj")}, i)

    R.load_calibs, R.allocate_global_sparsity, R.compress_nystrom, R.compress_qk, R.compress_vo = load_calibs, allocate, nystrom, qk, vo
    R._free = lambda: None
    rows = []
    real_rows = S._allgather_rows
    S._allgather_rows = lambda send, recv, counts, r: (rows.append((tuple(send.shape), list(counts))), real_rows(send, recv, counts, r))[1]
    real_load = torch.load

    def no_load(*a, **k):
        raise AssertionError("the gather must pack its send buffer from the held tensors, not read files back")
    chunk = list(range(n))
    mine = S.my_layers(chunk, rank, world)
    assert mine == [[0, 1, 2], [3, 4], []][rank]
    torch.load = no_load
    try:
        masks = R.compress_chunk(ad, ad.config, chunk, rank, world)
    finally:
        torch.load = real_load
    # the switches of the sharded calibration: only the last rank wants BI (and runs the whole forward); it owns no layer -> no hooks
    assert seen["calib"] == (mine, mine[-1] if mine else 0, rank == world - 1, not mine)
    assert not any(hasattr(ad, a) for a in ("calib_want_bi", "calib_stop_after", "calib_no_hooks"))      # (reset after the call)
    assert seen["bi"] == [0.5 + i for i in range(n)]                                             # everybody got rank 2's BI scores
    assert len(masks) == n and all(torch.equal(masks[i], _layer(i)[1]) for i in range(n))
    assert rows == [((max(len(mine), 1), rows[0][0][1]), [3, 2, 1])]                              # 3 + 2 + 1 rows on the wire, not 3 x 3
    assert ad.take_held_artifacts(0) == {}                                                       # (collected by the gather)
    S.finalize()
    if rank == 0:      # rank 0 holds every layer's files: its own from save_layer, the others' from the gathered records
        for i in chunk:
            got = {}
            for suffix in ("mlp", "qk", "vo"):
                got.update(torch.load(os.path.join(shared, f"layer_{i}_{suffix}")))
            assert all(torch.equal(got[k], v) for k, v in _layer(i)[0].items()), i


def test_three_ranks_balanced_partition_empty_last_rank_and_held_artefacts():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_balanced, args=(3, _free_port(), tmp), nprocs=3, join=True)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the script spawns two fresh ranks itself (here on CPU over gloo,
    plumbing only -- the kernels need a GPU), rank 0 prints the one JSON line, exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--plumbing-only"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["plumbing_only"] and out["gathered_layers"] == 6 and out["backend"] == "gloo"


def test_bench_launcher_propagates_a_failing_rank():
    """A child that dies must fail the whole command (here: the kernels' refusal to run without a GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "no GPU here" in p.stderr
