"""world_size-2 gloo test of the N>1 path: batch sharding is a disjoint cover, every rank's work is counted
once, the job time is the max over ranks, and no collective touches sample data."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    from mdgen_amd.sharding import shard_list, max_over_ranks, sum_over_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names = [f"pep{i}" for i in range(n_items)]
    mine = shard_list(names, rank, world)
    frames = 1000 * len(mine)                      # every peptide: one 1000-frame block
    dist.barrier()
    dt = max_over_ranks(0.1 * (rank + 1), dist)    # pretend rank r took 0.1*(r+1) s
    total = sum_over_ranks(frames, dist)
    dist.barrier()
    q.put((rank, mine, dt, total))
    dist.destroy_process_group()


def test_two_rank_batch_sharding():
    world, n_items = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(sum(shards, [])) == sorted(f"pep{i}" for i in range(n_items))     # disjoint cover
    assert abs(len(shards[0]) - len(shards[1])) <= 1
    assert all(abs(r[2] - 0.2) < 1e-9 for r in res)                                   # max over ranks
    assert all(r[3] == 1000 * n_items for r in res)                                   # whole-job frames


def test_shard_range_properties():
    from mdgen_amd.sharding import shard_range
    for n in (0, 1, 5, 16, 255, 256):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
