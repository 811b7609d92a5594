"""N > 1 host path on CPU: world_size-2 / 3 gloo jobs.  ONE connected body is sharded over the ranks (hot_set_comm +
hot_amd/dist.py); the engine is the CPU oracle (the HIP library needs a GPU: tests/test_gpu_multirank.py runs the same
comparison with it), the communicator, the shard split and the worker are the ones the GPU path uses."""
import numpy as np
import pytest

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


def test_irregular_body_three_ranks_gloo_oracle():
    """A hollow ball with a bar through it and a thinned half, cut into three shards: ragged colour blocks, uneven colours, coarse
    levels with gaps, rows near the cuts assembled from two ranks' particles."""
    from tests import multirank_worker as mw
    from tests.oracle_lib import load_oracle
    kw = dict(lsolver=3, levelCnt=3, max_iterations=3, cneps=1e-7)
    ranks = mw.launch(3, "oracle", -10, 1, kw)
    ref = mw.single(load_oracle(), -10, 1, kw)
    mw.compare(ranks, ref, 1e-11)


def test_one_body_three_ranks_whole_steps_gloo_oracle():
    """Two whole time steps (sort -> P2G -> solve to convergence -> G2P) on three ranks: same iteration counts, same particles."""
    from tests import multirank_worker as mw
    from tests.oracle_lib import load_oracle
    kw = dict(lsolver=3, levelCnt=2, cneps=1e-6)
    ranks = mw.launch(3, "oracle", 6, 1, kw, steps=2)
    ref = mw.single(load_oracle(), 6, 1, kw, steps=2)
    assert ranks[0]["iterations"] == ref["iterations"], (ranks[0]["iterations"], ref["iterations"])
    mw.compare(ranks, ref, 1e-9, tolp=1e-8, exact_counts=False)


def test_rank_local_gs_two_and_three_ranks_gloo_oracle():
    """hot_config.shard_gs = 1 (processor-block GS: a rank's coloured sweeps see only its own rows, one exchange per symmetric sweep
    instead of sixteen).  It is a different smoother, so the iterates are not the single-rank ones; what must hold: every rank still
    computes identical replicated data, the solve converges to the same minimiser (dv within the termination tolerance's reach),
    the number of L-BFGS iterations stays within 12 % of the colour-synchronous run on this small body (8^3 cells cut in 2 or 3: most nodes
    sit next to a cut; DESIGN.md §7 has the measured drift at larger sizes), and the collectives per V-cycle drop."""
    from tests import multirank_worker as mw
    from tests.oracle_lib import load_oracle
    kw = dict(lsolver=3, levelCnt=3, cneps=1e-10, max_iterations=400)
    ref = mw.single(load_oracle(), 8, 1, kw)
    exact = mw.launch(2, "oracle", 8, 1, kw)
    assert exact[0]["stats"]["iterations"] == ref["stats"]["iterations"] and ref["stats"]["converged"] == 1
    for world in (2, 3):
        ranks = mw.launch(world, "oracle", 8, 1, dict(kw, shard_gs=1))
        st = ranks[0]["stats"]
        for r in ranks[1:]:
            assert np.array_equal(r["dv"], ranks[0]["dv"]) and r["stats"]["iterations"] == st["iterations"]  # replicated decisions
        assert st["converged"] == 1
        drift = st["iterations"] - ref["stats"]["iterations"]
        print("rank-local GS, %d ranks: %d L-BFGS iterations (colour-synchronous / single rank: %d), collectives %s (colour-synchronous, 2 ranks: %s)"
              % (world, st["iterations"], ref["stats"]["iterations"], ranks[0]["comm_calls"], exact[0]["comm_calls"]))
        assert abs(drift) <= 0.12 * ref["stats"]["iterations"], (st["iterations"], ref["stats"]["iterations"])
        assert mw.rel(ranks[0]["dv"], ref["dv"]) < 1e-3, mw.rel(ranks[0]["dv"], ref["dv"])  # both stopped by the same test at cneps = 1e-10 (measured 7e-5; 5e-2 at cneps = 1e-7)
        if world == 2:  # (the CPU engine hands a colour over with an all-reduce)
            assert ranks[0]["comm_calls"]["allreduce"] < 0.5 * exact[0]["comm_calls"]["allreduce"]


def test_eight_ranks_small_subdomains_default_smoother_converges_gloo_oracle():
    """What `bench.py --gpus 8` runs by default since round 6, at the sub-domain size where the rank-local sweep fails (round 5: 24^3 cells per rank
    did not converge in 400 iterations): a 24^3-cell cube over EIGHT ranks — 12^3 cells each, every rank cut on three sides — with the default
    sharding knobs (colour-synchronous GS, ownership by the sweep).  One whole time step to convergence: the single-rank iteration count and
    particles, identical decisions on every rank.  (The rank-local sweep stays opt-in: hot_config.shard_gs = 1.)"""
    from tests import multirank_worker as mw
    from tests.oracle_lib import load_oracle
    kw = dict(lsolver=3, levelCnt=3, cneps=1e-6)
    ranks = mw.launch(8, "oracle", 24, 1, kw, steps=1, partition_min_rows=256, timeout=1500)
    ref = mw.single(load_oracle(), 24, 1, kw, steps=1)
    assert ref["stats"]["converged"] == 1 and all(o["stats"]["converged"] == 1 for o in ranks)
    assert all(o["iterations"] == ranks[0]["iterations"] for o in ranks)
    assert ranks[0]["iterations"] == ref["iterations"], (ranks[0]["iterations"], ref["iterations"])
    mw.compare(ranks, ref, 1e-8, tolp=1e-8, exact_counts=False)


def test_l1_scaled_rank_local_gs_converges_where_the_plain_one_does_not_gloo_oracle():
    """hot_config.shard_gs = 2 on the CPU engine: 16^3 cells over eight ranks (8^3 per rank).  The l1-scaled sweep converges in about the single-rank
    number of iterations (measured 41 against 43); the plain rank-local sweep (shard_gs = 1) is not run here — on this body it does not get there
    within the job's time-out."""
    from tests import multirank_worker as mw
    from tests.oracle_lib import load_oracle
    kw = dict(lsolver=3, levelCnt=3, cneps=1e-6, max_iterations=300)
    ranks = mw.launch(8, "oracle", 16, 1, dict(kw, shard_gs=2), steps=1, partition_min_rows=256, timeout=1500)
    ref = mw.single(load_oracle(), 16, 1, kw, steps=1)
    assert all(o["stats"]["converged"] == 1 for o in ranks) and ref["stats"]["converged"] == 1
    a, b = ranks[0]["iterations"][0], ref["iterations"][0]
    assert a <= 1.25 * b + 2, (a, b)
    for r in ranks[1:]:
        assert r["iterations"] == ranks[0]["iterations"]


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
