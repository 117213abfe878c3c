"""The N > 1 path: stream sharding and the benchmark's cross-rank bookkeeping, world_size 2 over gloo on CPU."""
import os
import socket

import pytest

from hevc_amd import sharding


def test_every_stream_has_exactly_one_owner():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            owned = [s for r in range(world) for s in sharding.streams_for_rank(n, r, world)]
            assert sorted(owned) == list(range(n))
    with pytest.raises(ValueError):
        sharding.streams_for_rank(4, 2, 2)


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.streams_for_rank(64, rank, world)
    elapsed = 1.0 + rank  # rank 1 is the slow one
    dist.barrier()
    tmax = sharding.max_over_ranks(elapsed, dist)
    total = sharding.sum_over_ranks(len(mine) * 60, dist)
    out[rank] = (tmax, total, mine[:3])
    dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][0] == out[1][0] == 2.0          # max over ranks
    assert out[0][1] == out[1][1] == 64 * 60      # whole-job frame count
    assert out[0][2] == [0, 2, 4] and out[1][2] == [1, 3, 5]
