"""Multi-GPU layout rehearsed on CPU: gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from buzzdetect_amd import sharding


def test_round_robin_partition_is_exact():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            mine = sharding.shard_indices(1000, r, world)
            assert all(sharding.owner_of(i, world) == r for i in mine)
            seen += mine
        assert sorted(seen) == list(range(1000))
    assert len(sharding.shard_indices(1000, 0, 8)) == 125          # config 4: 125 files per rank
    with pytest.raises(ValueError):
        sharding.shard_indices(10, 2, 2)


def test_interleave_inverts_sharding():
    items = [torch.full((1,), float(i)) for i in range(11)]
    per_rank = [[items[i] for i in sharding.shard_indices(11, r, 3)] for r in range(3)]
    back = sharding.interleave_round_robin(per_rank)
    assert [int(t.item()) for t in back] == list(range(11))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        files = sharding.shard_indices(5, rank, world)                 # 5 "files", ragged over 2 ranks
        rows = [torch.full((3 + f, 13), float(f)) for f in files]      # file f yields 3+f windows
        local = torch.cat(rows, 0)
        got = sharding.gather_rows(local, dst=0)
        if rank == 0:
            assert got is not None and len(got) == world
            per_rank = []
            for r, t in enumerate(got):
                fs = sharding.shard_indices(5, r, world)
                sizes = [3 + f for f in fs]
                assert t.shape == (sum(sizes), 13)
                per_rank.append(list(torch.split(t, sizes)))
            ordered = sharding.interleave_round_robin(per_rank)
            assert [int(t[0, 0].item()) for t in ordered] == [0, 1, 2, 3, 4]
            assert [t.shape[0] for t in ordered] == [3, 4, 5, 6, 7]
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_gather_rows_world_size_2_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_gather_rows_single_process_is_identity():
    t = torch.arange(26.0).reshape(2, 13)
    assert sharding.gather_rows(t)[0] is t
