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


# ---- FSQ aux loss under world > 1: the reference averages the batch-mean code distribution across ranks, in eval
# too (reference vidtok/modules/regularizers.py:49-59,240).  Each rank quantises its own shard; the host mirror
# (vidtok_amd/regularizers.py::_aux_stats) must produce the reference's per-rank aux_loss.  Runs the UNMODIFIED
# reference regularizer in the same 2-rank gloo job when /root/reference is present, the oracle formula otherwise.
def _fsq_worker(rank, world, port, q, use_reference):
    import pytest  # noqa: F401

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch_ops_ref as R
    import vidtok_amd.ops as ops
    from vidtok_amd.regularizers import FSQRegularizer

    for name in R.ALL:                      # CPU statements of the operator contracts (host-logic test)
        setattr(ops, name, getattr(R, name))
    ops._chk = lambda t, name: None
    levels = [8, 8, 8, 5, 5]
    kw = dict(levels=levels, entropy_loss_weight=0.1, commitment_loss_weight=0.25, diversity_gamma=1.0)
    g = torch.Generator().manual_seed(5)
    h_all = torch.randn((4, 5, 3, 6, 6), generator=g) * 0.8
    a, b = shard_range(4, world, rank)
    h = h_all[a:b].contiguous()
    z, log = FSQRegularizer(**kw)(h)
    if use_reference:
        from oracle.refload import _install_stubs

        _install_stubs()
        from vidtok.modules.regularizers import FSQRegularizer as RefFSQ

        zr, logr = RefFSQ(**kw).eval()(h)
        exp_aux, exp_idx = float(logr["aux_loss"]), logr["indices"]
    else:
        st, avg = R.fsq_aux_stats(h, levels, 100.0, return_avg=True)
        _, avg_all = R.fsq_aux_stats(h_all, levels, 100.0, return_avg=True)    # equal shards: mean of means
        exp_aux = float((st[0] - R.entropy(avg_all)) * 0.1 + st[2] * 0.25)
        exp_idx = R.fsq_quantize(h, levels)[1]
    st_local = R.fsq_aux_stats(h, levels, 100.0)
    local_aux = float((st_local[0] - st_local[1]) * 0.1 + st_local[2] * 0.25)
    q.put((rank, float(log["aux_loss"]), exp_aux, local_aux, bool(torch.equal(log["indices"], exp_idx.to(torch.int32)))))
    dist.barrier()
    dist.destroy_process_group()


def test_fsq_aux_loss_uses_cross_rank_mean():
    from oracle.refload import reference_available

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fsq_worker, args=(r, world, port, q, reference_available())) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, aux, exp, local, idx_ok in res:
        assert idx_ok
        assert abs(aux - exp) < 2e-5 * max(1.0, abs(exp)), (rank, aux, exp)
        assert abs(local - exp) > 1e-4, "test data must make the cross-rank mean matter"


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` outside a launcher must start 2 ranks by itself (the flag was a no-op).
    --selftest-spawn swaps the GPU work for the gloo metrics reduction, everything before it is the real launch path."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-spawn"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["clips"] == 8.0 and line["elapsed_s"] == 2.0
    # under a launcher with the wrong rank count the mismatch is an error, not a silent single-GPU run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-spawn"], env=env2,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_eight_rank_host_loop_stays_small():
    """8-GPU readiness without an 8-GPU box: eight ranks at once run the bench's real per-step host loop --
    AutoencodingEngine.forward in graph-replay mode through the real GraphedCall bookkeeping and the real KL regularizer with
    the reference's host-side noise draw (torch.randn of the CPU generator, distributions.py:16-18), each on its share of the
    host threads -- with every device operation stubbed (bench.py::selftest_host_loop).  The slowest rank's host time per
    step must stay under 4 ms (a step is ~75 ms of GPU time: the host then cannot pace a rank, and 8 ranks fit one box)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    best = None
    for attempt in range(2):        # a shared build host can hand eight 1-thread ranks a bad minute: the gate is on the better of two runs
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--selftest-spawn", "--selftest-steps", "40"], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        print(line)
        assert line["n_gpus"] == 8 and line["clips"] == 32.0 and line["steps"] == 40
        best = line["host_ms_per_step"] if best is None else min(best, line["host_ms_per_step"])
        if best < 4.0:
            break
    assert 0.0 < best < 4.0, best
