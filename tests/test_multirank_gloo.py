"""N > 1 host path on CPU: world_size-2 gloo job.  Each rank takes its body shard, advances it with the CPU oracle
(the GPU library needs a GPU; the sharding / clock / gather logic is identical), and the ranks check that the shards
partition the global scene, never share a grid node, and that the per-rank result equals the single-process result
for the same body (i.e. the by-body decomposition needs no data-path collective)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(n=5, ppc=8, E=5e4, nu=0.3, rho=2000.0, dtype=np.float64, levelCnt=2)


def _step(cloud):
    from tests.oracle_lib import load_oracle
    from hot_amd import synth
    ora = load_oracle()
    ctx = ora.context(dtype=1, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=2, cneps=1e-6)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    o, nrm = synth.sticky_floor(5.0, cloud["dx"])
    ctx.set_sticky_halfspaces(o, nrm)
    st = ctx.advance(1.0 / 24)
    return ctx.get_particles(), st, ctx.grid()["id2coord"]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hot_amd import parallel
    cloud = parallel.shard_cloud(CFG, rank, world)
    parts, st, coords = _step(cloud)
    # statistics travel through the process group exactly as in bench.py
    npart = parallel.sum_over_ranks(cloud["X"].shape[0], dist)
    tmax = parallel.max_over_ranks(st["ms_total"], dist)
    lo = torch.tensor([coords[:, 0].min(), coords[:, 0].max()], dtype=torch.int64)
    both = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(both, lo)
    q.put((rank, npart, tmax, [b.tolist() for b in both], parts["X"], st["iterations"]))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    from hot_amd import parallel
    world = 2
    assert parallel.assign_bodies(5, 2) == [(0, 3), (3, 5)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(parallel.shard_cloud(CFG, r, world)["X"].shape[0] for r in range(world))
    for rank, npart, tmax, ranges, X, its in res:
        assert npart == total
        assert tmax >= 0
        # grid node ranges of the two ranks do not overlap (no shared node => no halo)
        assert ranges[0][1] < ranges[1][0]
    # per-rank result == single-process result for the same body
    sys.path.insert(0, ROOT)
    ref_parts, ref_st, _ = _step(parallel.shard_cloud(CFG, 1, world))
    assert res[1][5] == ref_st["iterations"]
    assert np.array_equal(res[1][4], ref_parts["X"])
