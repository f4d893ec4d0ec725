"""world_size-2 gloo test of the N>1 path's only cross-rank logic: batch sharding and the metrics
reduction (max elapsed, summed counters) that bench.py performs over RCCL on GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vidtok_amd.sharding import reduce_metrics, shard_range


def test_shard_range_is_a_partition():
    for gb in (0, 1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = shard_range(gb, world, r)
                assert 0 <= a <= b <= gb and (b - a) in (gb // world, gb // world + 1)
                seen += list(range(a, b))
            assert seen == list(range(gb))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_range(9, world, rank)
    # each rank "processes" its clips: the per-clip result depends only on the clip (independent units)
    clips = torch.arange(9, dtype=torch.float64)[a:b]
    local = {"frames": 17.0 * (b - a), "checksum": float((clips * clips).sum())}
    out = reduce_metrics(1.0 + rank, local)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r]["elapsed_s"] == 2.0 and res[r]["world"] == 2          # max over ranks
        assert res[r]["frames"] == 17.0 * 9                                   # whole job
        assert res[r]["checksum"] == float(sum(i * i for i in range(9)))      # sharded == unsharded
