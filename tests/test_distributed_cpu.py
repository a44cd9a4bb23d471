"""world_size-2 (and 3) gloo tests on CPU for the N>1 path: utterance sharding + output all-gather.  The per-rank compute
is stood in for by the CPU oracle (tests may use it as the checker); the product's kernels need a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, batch: int, n: int, tmp: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from spectrograms_amd.distributed import gather_outputs, shard_range

        x = H.cfg2_batch(batch, n)
        p = orc.Params(n_fft=256, hop=64, n_mels=20, amp="db", floor_db=-80.0)
        start, count = shard_range(batch, world, rank)
        local = torch.from_numpy(orc.spectrogram_batch(p, x[start:start + count])) if count else \
            torch.empty((0, 20, orc.frame_count(n, 256, 64, True)), dtype=torch.float32)
        full = gather_outputs(local, batch, None)
        ref = orc.spectrogram_batch(p, x)
        assert full.shape == ref.shape, (full.shape, ref.shape)
        assert np.array_equal(full.numpy(), ref)
        # complex outputs travel as (re, im) pairs
        sp = orc.Params(n_fft=128, hop=32)
        ls = torch.view_as_real(torch.from_numpy(orc.stft_batch(sp, x[start:start + count]))).contiguous()
        fs = torch.view_as_complex(gather_outputs(ls, batch, None))
        assert np.array_equal(fs.numpy(), orc.stft_batch(sp, x))
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 6), (2, 5), (3, 7)])
def test_shard_and_gather_gloo(tmp_path, world, batch):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, batch, 3000, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _overlap_worker(rank: int, world: int, port: int, tmp: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spectrograms_amd.distributed import OverlappedGather

        og = OverlappedGather((4, 3), torch.float32, torch.device("cpu"), depth=2)
        shards = [torch.empty(4, 3) for _ in range(2)]  # the producer's rotating output buffers, as in bench.py
        seen = {}
        for i in range(7):
            og.wait_slot(i)                     # the collective that last read shards[i % 2] has finished
            shards[i % 2].fill_(100.0 * i + rank)   # "launch" step i
            dst = og.submit(i, shards[i % 2])
            seen[i] = dst
            if i >= 1:                          # step i - 1 may be consumed once its slot is waited for
                og.wait_slot(i - 1)
                for r in range(world):
                    assert torch.all(og.gathered[(i - 1) % 2][r] == 100.0 * (i - 1) + r)
        og.finish()
        for r in range(world):
            assert torch.all(seen[6][r] == 600.0 + r)
        open(os.path.join(tmp, f"ov{rank}"), "w").write("ok")
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_gather_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_overlap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ov{r}").exists() for r in range(world))


def test_shard_range_matches_reference_partition():
    from spectrograms_amd.distributed import shard_range

    assert [shard_range(8192, 8, r) for r in range(8)] == [(1024 * r, 1024) for r in range(8)]  # BASELINE configs[3]
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
