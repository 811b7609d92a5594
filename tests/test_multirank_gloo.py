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


# ---------------------------------------------------------------------------------------------------------------------
# One connected body over two ranks: hot_set_comm + hot_amd/dist.py over gloo, with the CPU oracle as the
# engine (the HIP library needs a GPU; tests/test_gpu_multirank.py runs the same comparison with it).
SHARD_CFGS = [
    dict(lsolver=3, levelCnt=3, max_iterations=5, cneps=1e-7),  # HOT: L-BFGS + 3-level Galerkin V-cycle, coloured GS partitioned on every level
    dict(lsolver=2, levelCnt=2, max_iterations=3, cneps=1e-7),  # projected Newton + MG-PCG (rebuilds the hierarchy every iteration)
    dict(lsolver=2, levelCnt=1, matrixFree=1, systemBCProject=0, max_iterations=3, cneps=1e-7),  # matrix-free product: scatter + all-reduce
]


@pytest.mark.parametrize("kw", SHARD_CFGS, ids=["lbfgs_mg3", "pn_mgpcg", "pn_matfree"])
def test_one_body_two_ranks_gloo_oracle(kw):
    """ONE cube split across two ranks reproduces the single-rank solve: node numbering bit-exact, dv after a fixed number of
    nonlinear iterations to round-off, identical iteration / line-search / V-cycle counts, particles after G2P."""
    from tests import multirank_worker as mw
    from tests.oracle_lib import load_oracle
    ranks = mw.launch(2, "oracle", 8, 1, kw)
    ref = mw.single(load_oracle(), 8, 1, kw)
    mw.compare(ranks, ref, 1e-11)
    assert ranks[0]["comm_calls"]["allreduce"] > 0 and ranks[0]["comm_calls"]["allgather"] > 0


def test_one_body_three_ranks_whole_steps_gloo_oracle():
    """Two whole time steps (sort -> P2G -> solve to convergence -> G2P) on three ranks: same iteration counts, same particles."""
    from tests import multirank_worker as mw
    from tests.oracle_lib import load_oracle
    kw = dict(lsolver=3, levelCnt=2, cneps=1e-6)
    ranks = mw.launch(3, "oracle", 6, 1, kw, steps=2)
    ref = mw.single(load_oracle(), 6, 1, kw, steps=2)
    assert ranks[0]["iterations"] == ref["iterations"], (ranks[0]["iterations"], ref["iterations"])
    mw.compare(ranks, ref, 1e-9, tolp=1e-8, exact_counts=False)


def test_shard_by_page_order_partitions_in_sort_order():
    from hot_amd import dist as hdist, synth
    from tests.oracle_lib import load_oracle
    for T in (np.float64, np.float32):
        c = synth.cube_cloud(7, ppc=8, dtype=T)
        ctx = load_oracle().context(dtype=1 if T == np.float64 else 0, dx=c["dx"])
        ctx.set_particles(c["X"], c["V"], c["mass"], c["vol"], c["mu"], c["lam"])
        ctx.sort()
        ix = ctx.indexing()
        assert np.array_equal(hdist.page_keys(c["X"], c["dx"], T), ix["particle_base_offset"] >> 12)  # the numpy keys are the library's
        shards = [hdist.shard_by_page_order(c, r, 3) for r in range(3)]
        pk = hdist.page_keys(c["X"], c["dx"], T)
        assert np.array_equal(np.sort(np.concatenate([s["index"] for s in shards])), np.arange(len(pk)))
        for a, b in zip(shards, shards[1:]):
            assert pk[a["index"]].max() < pk[b["index"]].min()  # contiguous page ranges, whole pages only
        sizes = [len(s["index"]) for s in shards]
        per_page = np.unique(pk, return_counts=True)[1].max()
        assert max(sizes) - min(sizes) <= 2 * per_page  # balanced to within the granularity of whole pages
