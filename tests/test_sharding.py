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


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (fresh processes), shards the streams with
    streams_for_rank and prints one line from rank 0.  --rehearse: the plumbing only, no GPU needed."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--total-streams", "7", "--rehearse"], cwd=root, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode("latin1")
    assert out.returncode == 0, text[-2000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["streams_total"] == 7 and d["streams_rank0"] == [0, 2, 4, 6] and d["max_over_ranks"] == 2.0
