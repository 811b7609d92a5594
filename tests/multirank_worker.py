"""Worker of the multi-rank tests: one process per rank, one connected body sharded over the ranks (hot_amd/dist.py).
`libkind` selects the library: "oracle" (CPU restatement, host pointers, the N > 1 host path exercised on CPU with gloo) or
"hip" (the product on a GPU; on the single-GPU test box all ranks share device 0 and talk through gloo, on a multi-GPU node
each rank takes its own device and RCCL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scene(n, dtype, cells=None):
    """n > 0: a solid cube of n^3 cells.  n < 0: an irregular body carved out of a |n|^3 cube — a hollow ball with a bar through it and
    random holes (cells with few or no particles, ragged colour blocks, coarse levels with gaps), 8 or 2 particles per cell by octant."""
    from hot_amd import synth
    c = synth.cube_cloud(abs(n), ppc=8, dtype=np.float64 if dtype == 1 else np.float32, cells=cells)
    if n > 0:
        return c
    X = c["X"].astype(np.float64)
    ctr = X.mean(0)
    r = np.linalg.norm(X - ctr, axis=1)
    ext = (X.max(0) - X.min(0)).min()
    keep = ((r < 0.48 * ext) & (r > 0.2 * ext)) | ((np.abs(X[:, 0] - ctr[0]) < 0.012) & (np.abs(X[:, 1] - ctr[1]) < 0.012))
    rng = np.random.default_rng(17)
    keep &= (rng.random(len(X)) < 0.25) | (X[:, 2] > ctr[2])  # thinned lower half
    return {k: (v[keep] if isinstance(v, np.ndarray) else v) for k, v in c.items()}


def run_case(lib, cloud, comm, cfgkw, steps, dt=1.0 / 24):
    """A fixed number of nonlinear iterations of one step (or whole steps) on this rank's shard; returns replicated grid data
    and the shard's particles."""
    from hot_amd import synth
    ctx = lib.context(dx=cloud["dx"], gravity=(0, -9.8, 0), **cfgkw)
    if comm is not None:
        ctx.set_comm(comm)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    if comm is not None:
        ctx.set_particle_ids(cloud["index"])  # positions in the whole body: sort-key tie break, identity of a migrating particle
    o, nrm = synth.sticky_floor(5.0, cloud["dx"])
    ctx.set_sticky_halfspaces(o, nrm)
    out = {}
    if steps == 0:  # pieces: one solve with the iteration cap of cfgkw, then G2P
        ctx.sort(), ctx.p2g(), ctx.begin_step(dt)
        g = ctx.grid()
        out["id2coord"], out["mass"], out["v"] = g["id2coord"], g["mass"], g["v"]
        out["e0"] = ctx.update_state(ctx.get_dv())
        out["r0"] = ctx.residual()
        st = ctx.solve()
        out["dv"] = ctx.get_dv()
        out["stats"] = st
        x = np.random.default_rng(5).standard_normal((ctx.Nn, 3))
        if cfgkw.get("matrixFree"):  # no assembled matrix / hierarchy: the operator itself
            out["vcycle"] = out["spmv"] = ctx.matfree_multiply(x)
            out["levels"] = []
        else:
            out["vcycle"] = ctx.vcycle(ctx.project(x))
            out["spmv"] = ctx.spmv(0, x)
            out["levels"] = [ctx.level(l)["id2coord"] for l in range(st["num_levels"])]
        ctx.g2p(dt)
    else:
        sts = [ctx.advance(dt) for _ in range(steps)]
        out["stats"] = sts[-1]
        out["iterations"] = [s["iterations"] for s in sts]
    if cfgkw.get("profile"):
        out["profile"] = ctx.profile()
    out["particles"] = ctx.get_particles()
    out["ids"] = ctx.particle_ids() if comm is not None else np.arange(len(out["particles"]["X"]), dtype=np.int32)
    return out


def worker(rank, world, port, q, libkind, n, dtype, cfgkw, steps, backend, partition_min_rows, shard_ids=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    try:
        import torch
        import torch.distributed as dist
        from hot_amd import dist as hdist
        device = None
        if libkind == "hip":
            import hot_amd
            ndev = torch.cuda.device_count()
            dev = rank % ndev
            torch.cuda.set_device(dev)
            device = torch.device("cuda", dev)
            lib = hot_amd.load()
            cfgkw = dict(cfgkw, device=dev)
        else:
            from tests.oracle_lib import load_oracle
            lib = load_oracle()
        dist.init_process_group(backend, rank=rank, world_size=world)
        comm = hdist.TorchComm(device=device, partition_min_rows=partition_min_rows)
        cloud = scene(n, dtype)
        if shard_ids is not None:  # the caller's partition (e.g. the one another run's migration ended with): global particle ids per rank
            sel = np.sort(np.asarray(shard_ids[rank]))
            shard = {k: (v[sel] if isinstance(v, np.ndarray) and len(v) == len(cloud["X"]) else v) for k, v in cloud.items()}
            shard["index"] = sel.astype(np.int32)
        else:
            shard = hdist.shard_by_page_order(cloud, rank, world)
        out = run_case(lib, shard, comm, dict(cfgkw, dtype=dtype), steps)
        out["comm_calls"] = dict(comm.calls)
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent instead of hanging it
        import traceback
        q.put((rank, dict(error=traceback.format_exc())))
        raise


def launch(world, libkind, n, dtype, cfgkw, steps=0, backend="gloo", partition_min_rows=1, timeout=900, shard_ids=None, attempt=0):
    """One job of `world` processes.  A rendezvous that fails (the TCP store's port taken, a rank that starts too late on a loaded box: c10d's
    DistNetworkError) is retried twice on another port; any other failure of a rank is raised."""
    import torch.multiprocessing as mp
    try:
        return _launch(mp, world, libkind, n, dtype, cfgkw, steps, backend, partition_min_rows, timeout, shard_ids, attempt)
    except RuntimeError as e:
        if attempt < 2 and ("DistNetworkError" in str(e) or "DistStoreError" in str(e) or "Address already in use" in str(e)):
            return launch(world, libkind, n, dtype, cfgkw, steps, backend, partition_min_rows, timeout, shard_ids, attempt + 1)
        raise


def _launch(mp, world, libkind, n, dtype, cfgkw, steps, backend, partition_min_rows, timeout, shard_ids, attempt):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + n + 977 * attempt) % 2000
    procs = [ctx.Process(target=worker, args=(r, world, port, q, libkind, n, dtype, cfgkw, steps, backend, partition_min_rows, shard_ids)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=timeout)
            if "error" in out:
                raise RuntimeError(f"rank {r} failed:\n{out['error']}")
            res[r] = out
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for p in procs:
        if p.exitcode != 0:
            raise RuntimeError(f"a rank exited with code {p.exitcode}")
    return [res[r] for r in range(world)]


def hdist_initial(n, dtype, rank, world):
    """global ids of the shard rank `rank` starts with"""
    from hot_amd import dist as hdist
    return hdist.shard_by_page_order(scene(n, dtype), rank, world)["index"]


def single(lib, n, dtype, cfgkw, steps=0):
    cloud = scene(n, dtype)
    return run_case(lib, cloud, None, dict(cfgkw, dtype=dtype), steps)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def compare(ranks, ref, tol, tolp=None, exact_counts=True):
    """Every rank holds the replicated grid data of the whole body: identical across ranks and equal to the single-rank run;
    the shards' particles put back in place equal the single-rank particles."""
    tolp = tol if tolp is None else tolp
    world = len(ranks)
    for r in range(world):
        o = ranks[r]
        if "id2coord" in o:
            assert np.array_equal(o["id2coord"], ref["id2coord"]), "node numbering differs from the single-rank run"
            for l, (a, b) in enumerate(zip(o["levels"], ref["levels"])):
                assert np.array_equal(a, b), f"level {l} numbering differs"
            assert rel(o["mass"], ref["mass"]) < tol and rel(o["v"], ref["v"]) < tol * 10
            assert abs(o["e0"] - ref["e0"]) < tol * max(abs(ref["e0"]), 1e-12)
            assert rel(o["r0"], ref["r0"]) < tol * 10
            assert rel(o["spmv"], ref["spmv"]) < tol * 10, rel(o["spmv"], ref["spmv"])
            assert rel(o["vcycle"], ref["vcycle"]) < tol * 100, rel(o["vcycle"], ref["vcycle"])
            assert rel(o["dv"], ref["dv"]) < tol * 100, rel(o["dv"], ref["dv"])
            if r > 0:  # replicated data is bit-identical on every rank: that is what keeps the ranks' control flow in step
                assert np.array_equal(o["dv"], ranks[0]["dv"]) and np.array_equal(o["vcycle"], ranks[0]["vcycle"])
        if exact_counts:
            for k in ("iterations", "vcycles", "linesearch_trials", "linear_iterations", "dropped_pairs", "num_nodes", "num_levels"):
                assert o["stats"][k] == ref["stats"][k], (k, o["stats"], ref["stats"])
        assert abs(o["stats"]["energy"] - ref["stats"]["energy"]) < tol * 100 * max(abs(ref["stats"]["energy"]), 1e-12)
    idx = np.concatenate([o["ids"] for o in ranks])  # the ranks' CURRENT particles (they migrate with the page ranges)
    assert np.array_equal(np.sort(idx), np.arange(len(ref["particles"]["X"])))  # the shards partition the body
    for o in ranks:
        assert np.all(np.diff(o["ids"]) > 0)  # returned in ascending id order
    for k in ("X", "V", "F", "C"):
        got = np.concatenate([o["particles"][k] for o in ranks])
        assert rel(got, ref["particles"][k][idx]) < tolp * (1 if k == "X" else 100), (k, rel(got, ref["particles"][k][idx]))
