"""CPU suite, part 3: the N>1 path of bench.py (pair sharding + timing reduction) under
torch.distributed with the gloo backend, world_size 2."""
import os

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import faldoi_ipol_amd as F


def test_assign_pairs_round_robin_covers_every_pair_once():
    for world in (1, 2, 3, 8):
        got = sorted(p for r in range(world) for p in F.shard.assign_pairs(32, world, r))
        assert got == list(range(32))
    assert F.shard.assign_pairs(32, 8, 3) == [3, 11, 19, 27]
    with pytest.raises(ValueError):
        F.shard.assign_pairs(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import faldoi_ipol_amd as F2
    mine = F2.shard.assign_pairs(5, world, rank)
    secs = 1.0 + rank                  # rank 1 is the slow one
    t, u = F2.shard.reduce_timing(secs, 100.0 * len(mine), dist)
    dist.barrier()
    q.put((rank, t, u, mine))
    dist.destroy_process_group()


def test_reduce_timing_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, u, mine in res:
        assert t == 2.0 and u == 500.0          # MAX of seconds, SUM of units on every rank
    assert res[0][3] == [0, 2, 4] and res[1][3] == [1, 3]
