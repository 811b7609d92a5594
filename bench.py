#!/usr/bin/env python3
"""bench.py — HOT hot path on MI355X: one "step" = one backward-Euler time step (sort -> P2G -> L-BFGS/Galerkin-MG
solve -> G2P) of the BASELINE.json config-2 stand-in (63^3-cell cube, 8 ppc = 2.0 M particles,
fp64, 3 MG levels, -lsolver 3 -smoother 5 -coarseSolver 2 --project --linesearch --bcproject --usecn).

Prints ONE JSON line on rank 0.  `value` = ms per nonlinear (L-BFGS) iteration, Hessian + hierarchy build excluded
(reported separately and amortised); extra keys carry the P2G+G2P Mparticles/s half of the metric.
`roofline` = the kernel (device symbol) with the largest share of the timed region, from HIP events recorded on the
library's launch stream (hot_config.profile); its `avg_launch_ms` is what rocprofv3 --kernel-trace --stats reports
for the same symbol (profiles/).  `cpu_baseline` = the CPU oracle (a port of the reference's TBB decomposition to
OpenMP), REBUILT ON THE BENCH HOST (-O3 -march=native there, like the reference's CMakeLists.txt:28) and timed on its cores on the
headline configuration itself: the first --cpu-iters L-BFGS iterations (default 12) of the first time step, Hessian + hierarchy build
timed separately, in each of two variants ("faithful": serial where the reference is serial; "fair": those sections parallelised as
well); the GPU takes the same bounded step beside it.  ~15 s of CPU work per variant on a 16-core host.

N > 1: one process per GPU (launched by torch.distributed.run, or spawned here when WORLD_SIZE is unset), RCCL through
torch.distributed.  ONE connected body is sharded over the ranks (hot_set_comm, hot_amd/dist.py): particle
ranges of the global sort order per rank, node tiles summed between the ranks that share a block, matrix rows owned by one rank
each, DOF vectors on owned rows + halos, inner products all-reduced in batches, coloured Gauss-Seidel colour-synchronous across the ranks (the
reference's update order and the single-rank iterates, sixteen halo exchanges per symmetric sweep; --shard-gs 1 = rank-local / processor-block sweeps with one exchange, opt-in).  Weak scaling: the body grows so that every GPU keeps C2's particle count
(N = 8: a 126^3-cell body of 16 M particles, BASELINE config 4's size).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)

SYMBOL = {  # profile-record prefix -> device symbol as rocprofv3 names it
    "gs_forward": "hot::k_gs_subst<T,true,8>", "gs_backward": "hot::k_gs_subst<T,false,8>", "spmv": "hot::k_spmv<T>",
    "gs_forward_off": "hot::k_gs_offblock<T>", "gs_backward_off": "hot::k_gs_offblock<T>",
    "gs_forward_fused": "hot::k_gs_colour<T,", "gs_backward_fused": "hot::k_gs_colour<T,",  # <T, true, D>, <T, false, D> and <T, true, D, true> (the turn): one kernel, two sweep directions
    "gs_forward_chained": "hot::k_gs_sweep<T,true,SB>", "gs_backward_chained": "hot::k_gs_sweep<T,false,SB>",
    "gs_residual": "hot::k_gs_residual<T>", "hessian_assemble": "hot::k_hessian_rows<T>", "state_update": "hot::k_state<T>",
    "force_scatter": "hot::k_force_cells2<T>", "p2g": "hot::k_p2g_stream<T,true>", "g2p": "hot::k_g2p<T,0,true>",
}


def pmc_traffic(symbol, dtype_name):
    """HBM bytes per launch of `symbol` from the newest committed PMC summary (profiles/*_pmc_summary.json, produced
    by profiles/run_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH_SIZE doubled
    for gfx950).  PMC counters cannot be collected from inside the timed process, so this is a recorded value; None
    if there is no summary for this symbol."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        table = json.load(fh)
    stamp = table.get("_build", {}).get("source_sha16")
    if stamp is not None:  # the summary names the sources it was measured on (profiles/summarize_pmc.py)
        import hashlib
        h = hashlib.sha256()
        for f in sorted(glob.glob(os.path.join(here, "hot_amd", "csrc", "*.hip")) + glob.glob(os.path.join(here, "hot_amd", "csrc", "*.h"))):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
        stale = h.hexdigest()[:16] != stamp
    else:  # older summaries: file times
        stale = max((os.path.getmtime(f) for f in glob.glob(os.path.join(here, "hot_amd", "csrc", "*.hip"))), default=0.0) > os.path.getmtime(files[-1])
    if stale:
        print("bench: %s was measured on other kernel sources than hot_amd/csrc now holds: roofline.traffic is a RECORDED value of an earlier build (re-run profiles/run_profiles.sh)" % os.path.basename(files[-1]), file=sys.stderr)
    want = symbol.replace("<T", "<" + dtype_name).replace(" ", "")
    tot, n = 0.0, 0.0
    for name, rec in table.items():  # (a symbol prefix names all its instantiations: average over their launches)
        if want in name.replace(" ", "") and "hbm_bytes_per_launch" in rec:
            k = float(rec.get("FETCH_SIZE", {}).get("n", 1) or 1)
            tot += rec["hbm_bytes_per_launch"] * k
            n += k
    return (tot / n, os.path.relpath(files[-1], here)) if n else (None, None)


def host_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def make_ctx(lib, cloud, cfg, device=0, profile=0, comm=None, **over):
    from hot_amd import synth
    kw = dict(dtype=1 if cloud["X"].dtype == np.float64 else 0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=cfg["levelCnt"], device=device, profile=profile)
    kw.update(synth.plasticity_kwargs(cfg))
    kw.update(over)
    ctx = lib.context(**kw)
    if isinstance(comm, str):  # "rccl": the library's native communicator on the context's own stream (hot_amd/csrc/rccl_comm.hip)
        from hot_amd import dist as hdist
        import torch
        import torch.distributed as tdist
        ok, why = 1, None
        try:
            hdist.attach_rccl(ctx)
            ctx.rccl_selftest()  # every callback once with known data across the group (device and host payloads, ring exchange)
        except Exception as e:  # RCCL not loadable / not attachable / a collective returned a wrong result
            ok, why = 0, e
        flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", device))
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)  # the choice of communicator is the group's, not a rank's
        if int(flag.item()) == 0:  # torch.distributed collectives instead, on every rank
            print("bench: native RCCL communicator not used (%r on this rank), using TorchComm" % (why,), file=sys.stderr)
            ctx._fallback_comm = hdist.TorchComm(device=torch.device("cuda", device))
            ctx.set_comm(ctx._fallback_comm)
        elif tdist.get_rank() == 0:
            print("bench: pre-flight: hot_rccl_selftest passed on all %d ranks (all-gather, all-reduce, personalised Send / Recv exchange with checked contents); communicator = native RCCL on the context's stream"
                  % tdist.get_world_size(), file=sys.stderr)
    elif comm is not None:
        ctx.set_comm(comm)  # this rank's shard of ONE body (hot_amd/dist.py)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    if comm is not None and cloud.get("index") is not None:
        ctx.set_particle_ids(np.asarray(cloud["index"], np.int32))  # positions in the whole body: sort-key tie break, identity of a migrating particle
    o, n = synth.sticky_floor(cloud["corner"][1], cloud["dx"])
    ctx.set_sticky_halfspaces(o, n)
    return ctx


def algorithmic_bytes(name, s, Np, levels, launches_per_half_sweep=8.0):
    """SURVEY.md §8(d) per-launch algorithmic (compulsory) bytes of one profile record; None if not modelled."""
    base, _, lv = name.rpartition("_L")
    if base in ("gs_forward_off", "gs_backward_off", "gs_forward", "gs_backward", "gs_forward_fused", "gs_backward_fused") and lv.isdigit() and len(levels[int(lv)]) > 2 and levels[int(lv)][2] is not None:
        # the finest level's colour pass is two kernels (one launch each per colour): k_gs_offblock streams the off-block half rows
        # (values + tagged column ids, the gathered x is cache traffic), reads rhs and the 32-byte row record, writes rhs - sum;
        # k_gs_subst reads the premultiplied in-block couplings (no column ids), D^-1 (forward also D) and p1, writes x and hD
        N, nnzb, inb = levels[int(lv)]
        if base.endswith("_fused"):  # both of the above in ONE launch per colour (k_gs_colour); the previous-colour sums stay in LDS
            fw = base == "gs_forward_fused"
            return ((nnzb - N - inb) / 2.0 * (9 * s + 4) + N * (6 * s + 32) + inb / 2.0 * 9 * s + N * ((18 if fw else 9) * s + 3 * s + 6 * s + 4)) / launches_per_half_sweep
        if base.endswith("_off"):
            return ((nnzb - N - inb) / 2.0 * (9 * s + 4) + N * (6 * s + 32)) / launches_per_half_sweep
        return (inb / 2.0 * 9 * s + N * ((18 if base == "gs_forward" else 9) * s + 3 * s + 6 * s + 4)) / launches_per_half_sweep
    if base in ("spmv", "gs_forward", "gs_backward", "gs_residual") and lv.isdigit():
        N, nnzb = levels[int(lv)][:2]
        off = max(nnzb - N, 0) / 2.0  # blocks strictly preceding (or following) the row in the sweep order
        if base == "spmv":
            return nnzb * (9 * s + 4) + N * 6 * s
        if base == "gs_residual":
            return off * (9 * s + 4) + N * 9 * s
        per_half_sweep = off * (9 * s + 4) + N * ((18 if base == "gs_forward" else 9) * s + 6 * s)
        return per_half_sweep / launches_per_half_sweep  # one launch per (colour, sub-block)
    Nn = levels[0][0]
    if name == "p2g":
        return Np * 16 * s + Nn * 4 * s
    if name == "g2p":
        return Nn * 3 * s + Np * (3 * s + 24 * s) + Np * 18 * s  # + F in/out: evolveStrain is fused into the kernel
    if name == "state_update":
        return Np * (3 + 9 + 3 + 9 + 9) * s + Nn * 3 * s
    if name == "force_scatter":
        return Np * 12 * s + Nn * 3 * s
    if name == "hessian_assemble":
        return Np * (45 + 12) * s + levels[0][1] * 9 * s
    return None


def measured_copy_bandwidth(device):
    """GB/s of a device-to-device copy on this GPU, bytes read + bytes written (the attainable streaming rate next to the 8 TB/s of the data sheet)."""
    import torch
    n = 1 << 27  # doubles: 1 GiB per buffer
    a = torch.empty(n, dtype=torch.float64, device=torch.device("cuda", device))
    b = torch.ones(n, dtype=torch.float64, device=torch.device("cuda", device))
    for _ in range(3):
        a.copy_(b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    reps = 20
    for _ in range(reps):
        a.copy_(b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * n * 8 / (ms * 1e-3) / 1e9


def rocprof_avg_ms(symbol, tname):
    """Average duration (ms) of `symbol` in the newest committed rocprofv3 kernel-stats summary (profiles/rNN_kernel_stats.csv), or None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats.csv")))
    if not files:
        return None, None
    want = symbol.replace("<T", "<" + tname).split("<")[0]
    tot_ns, calls = 0.0, 0
    try:
        for row in csv.DictReader(open(files[-1])):
            name = row.get("Name") or row.get("KernelName") or ""
            if want in name and ("<" + tname) in name:
                c = float(row.get("Calls") or 0)
                tot_ns += float(row.get("TotalDurationNs") or 0) if row.get("TotalDurationNs") else float(row.get("AverageNs") or 0) * c
                calls += c
    except Exception:
        return None, None
    return (tot_ns / calls * 1e-6, os.path.basename(files[-1])) if calls else (None, None)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--cells", type=int, default=0, help="override the cube edge (cells) of the per-GPU body for quick runs")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-cells", type=int, default=0, help="cube edge of the CPU baseline's sample; 0 = the benchmark configuration itself")
    ap.add_argument("--cpu-iters", type=int, default=12, help="L-BFGS iterations of the CPU baseline's bounded sample (0 = the whole first time step, ~70 s per variant at C2)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N > 1 run (nccl = RCCL; gloo with --share-gpu on a one-GPU box)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"], help="N > 1: native stream-ordered RCCL communicator of the library (falls back to torch if RCCL "
                    "cannot be attached) or hot_amd.dist.TorchComm (torch.distributed collectives, host-synchronous)")
    ap.add_argument("--shard-gs", type=int, default=0, choices=[0, 1, 2], help="N > 1, coloured GS across ranks: 0 = colour-synchronous (default since round 6: the reference's update order, the single-rank "
                    "iterates and iteration counts, sixteen halo exchanges per symmetric sweep), 1 = processor-block / rank-local (one exchange per symmetric sweep; another smoother: "
                    "iteration counts drift by +-15 - 25 % and small sub-domains — 24^3 cells per rank — do not converge, profiles/r05_shard_ownership.txt), "
                    "2 = rank-local with the l1 norms of a row's off-rank couplings added to its diagonal block: convergent whatever the sub-domain size")
    ap.add_argument("--shard-owner", type=int, default=0, choices=[0, 1, 2], help="N > 1, hot_config.shard_owner: 0 = by the smoother (default: page-range ownership under --shard-gs 0, first touch under --shard-gs 1), "
                    "1 = the first-touching rank owns a block (rounds 2 - 4), 2 = the rank whose page range holds the block (balanced along the cuts)")
    ap.add_argument("--watchdog-s", type=float, default=1500.0, help="N > 1: abort the rank (exit code 3) if the run has not finished after this many seconds (a peer that died or a wedged collective would otherwise hang the job); 0 = off")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="N > 1: weak = the body grows so that every GPU keeps the configuration's particle count (default); "
                    "strong = the configuration's own body (e.g. --config C4 --gpus 4: BASELINE's 16 M particles over four GPUs), every rank generating only its cell planes")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on device 0 (functional check of the N > 1 path on a one-GPU box, not a measurement)")
    return ap.parse_args()


def _spawned(rank, world, port, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.argv = argv
    main()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: start the ranks ourselves, one process per GPU
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(args.gpus, 29400 + os.getpid() % 500, sys.argv), nprocs=args.gpus, join=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu and args.backend == "nccl":
        args.backend = "gloo"  # RCCL refuses two ranks on one device
    cores = host_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(cores, 32))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    import torch
    import hot_amd
    from hot_amd import parallel, synth

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP library has no CPU fallback)"
    torch.cuda.set_device(local)
    dist = None
    comm = None
    if world > 1 and args.watchdog_s > 0:
        import threading

        def _abort():
            print("bench: rank %d still running after %.0f s, aborting" % (rank, args.watchdog_s), file=sys.stderr, flush=True)
            os._exit(3)

        wd = threading.Timer(args.watchdog_s, _abort)
        wd.daemon = True
        wd.start()
    if world > 1:
        import torch.distributed as dist_
        from hot_amd import dist as hdist
        dist = dist_
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
        comm = hdist.TorchComm(device=torch.device("cuda", local)) if (args.comm == "torch" or args.backend != "nccl") else "rccl"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lib = hot_amd.load()
    cfg = dict(synth.CONFIGS[args.config])
    n1 = args.cells or cfg["n"]  # per-GPU body edge
    if args.scaling == "strong":
        n = n1  # the configuration's own body whatever the number of ranks
        cloud = parallel.slab_cloud(cfg, rank, world, n=n)
    else:
        n = parallel.cells_for_world(n1, world)  # edge of the one body all ranks share
        cloud = parallel.shard_cloud(cfg, rank, world, n=n)
    Np = cloud["X"].shape[0]
    s = 8 if cfg["dtype"] == np.float64 else 4
    dt = cfg["dt"]

    ctx = make_ctx(lib, cloud, cfg, device=local, comm=comm, **(dict(shard_gs=args.shard_gs, shard_owner=args.shard_owner) if world > 1 else {}))
    for _ in range(args.warmup):
        ctx.advance(dt)
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(ctx.advance(dt))
    ctx.sync()
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dist, "cuda")

    iters = sum(st["iterations"] for st in stats)
    build_ms = sum(st["ms_hessian"] + st["ms_mg_build"] for st in stats)
    solve_ms = sum(st["ms_solve"] for st in stats)
    ms_per_iter = parallel.max_over_ranks((solve_ms - build_ms) / max(iters, 1), dist, "cuda")
    total_particles = parallel.sum_over_ranks(Np, dist, "cuda")

    # ---- profiled pass (HIP events on the launch stream) for the roofline and the transfer half of the metric
    roof, transfers, prof_top = None, None, None
    used_fallback_comm = hasattr(ctx, "_fallback_comm")
    if world == 1:
        del ctx  # the timed context's device memory goes before the profiled one is created (C5's whole body: two of them do not fit 288 GB)
    if rank == 0 and world == 1:
        pctx = make_ctx(lib, cloud, cfg, device=local, profile=1)
        pctx.advance(dt)
        pctx.profile_reset()
        # the transfer kernels run once per step and show an occasional 2 x outlier under event timing: five steps where a step is short (C1 - C3)
        nprof = 5 if elapsed / max(args.steps, 1) < 1.5 else max(1, min(2, args.steps))
        pst, table, xfer_steps = [], {}, []
        for _ in range(nprof):  # one table per step: the transfer kernels run once a step, their time is reported as the median over the steps
            pctx.profile_reset()
            pst.append(pctx.advance(dt))
            t1 = pctx.profile()
            for k, v in t1.items():
                r = table.setdefault(k, dict(calls=0, total_ms=0.0))
                r["calls"] += v["calls"]
                r["total_ms"] += v["total_ms"]
            xfer_steps.append({k: t1[k]["total_ms"] / t1[k]["calls"] for k in ("p2g", "p2g_reduce", "g2p") if k in t1})
        levels = [(pctx.level(l, coords=False)["nrows"], pctx.level_nnzb(l), pctx.level_inblock_nnzb(l) if ("gs_forward_off_L%d" % l) in table or ("gs_forward_fused_L%d" % l) in table else None) for l in range(pst[-1]["num_levels"])]
        total_ms = sum(v["total_ms"] for v in table.values())
        groups = {}
        for name, rec in table.items():
            base = name.rpartition("_L")[0] if name.rpartition("_L")[2].isdigit() else name
            sweeps = table.get("gs_symsweeps_L" + name.rpartition("_L")[2])
            lph = rec["calls"] / sweeps["calls"] if sweeps and base in ("gs_forward", "gs_backward", "gs_forward_off", "gs_backward_off", "gs_forward_fused", "gs_backward_fused") else 8.0  # launches per half sweep
            if base in ("gs_forward", "gs_backward") and lph < 1.5:
                base += "_chained"  # coarse levels: the whole half sweep is one k_gs_sweep launch (a different device symbol)
            g = groups.setdefault(SYMBOL.get(base, base), dict(ms=0.0, calls=0, bytes=0.0, modelled=True, records={}))  # by device symbol: the forward and backward off-block sums are one kernel
            g["ms"] += rec["total_ms"]
            g["calls"] += rec["calls"]
            ab = algorithmic_bytes(name, s, Np, levels, lph)
            if ab is None:
                g["modelled"] = False
            else:
                g["bytes"] += ab * rec["calls"]
            g["records"][name] = dict(calls=rec["calls"], avg_ms=rec["total_ms"] / rec["calls"], algorithmic_bytes_per_launch=ab)
        top = max((k for k in groups if groups[k]["modelled"]), key=lambda k: groups[k]["ms"])
        g = groups[top]
        avg_ms = g["ms"] / g["calls"]
        achieved = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(top, "double" if s == 8 else "float") if args.config == "C2" and not args.cells else (None, None)
        copy_torch_gbs = measured_copy_bandwidth(local)
        copy_gbs = pctx.copy_bandwidth(1 << 30, 20)  # the library's own 16-byte-per-lane copy kernel on its stream (hot_copy_bandwidth)
        rp_avg_ms, rp_src = rocprof_avg_ms(top, "double" if s == 8 else "float") if args.config == "C2" and not args.cells else (None, None)
        roof = {"kernel": top, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "timer": "HIP events around every launch on the library's stream, averaged over the launches of the profiled steps (includes the launch boundary, 3 - 4 us per launch)",
                # the attainable figure, measured on this box in this run: a device-to-device copy of 1 GiB (read + write bytes counted)
                "peak_measured": copy_gbs, "peak_measured_how": "hot_copy_bandwidth: the library's own copy kernel (one non-temporal 16-byte piece per thread) over 1 GiB, 20 launches between two HIP events on the library's stream, bytes read + bytes written",
                "peak_measured_torch": copy_torch_gbs,  # torch Tensor.copy_ of 1 GiB: the figure of rounds 4 - 5
                "frac_of_measured": achieved / copy_gbs if copy_gbs else None,
                # the kernel-only duration of the committed rocprofv3 --kernel-trace --stats summary of this command (no launch boundary), for comparison
                "avg_launch_ms_rocprofv3": rp_avg_ms, "frac_rocprofv3": (g["bytes"] / g["calls"]) / (rp_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rp_avg_ms else None, "rocprofv3_source": rp_src,
                "levels": [{"rows": int(l[0]), "nnzb": int(l[1]), "nnzb_in_block": (int(l[2]) if l[2] is not None else None)} for l in levels],
                "avg_launch_ms": avg_ms, "launches": g["calls"], "algorithmic_bytes_per_launch": g["bytes"] / g["calls"], "share_of_kernel_time": g["ms"] / total_ms,
                "per_level": g["records"]}
        med = lambda xs: sorted(xs)[len(xs) // 2] if len(xs) % 2 else 0.5 * (sorted(xs)[len(xs) // 2 - 1] + sorted(xs)[len(xs) // 2])
        t_p2g = med([x["p2g"] + x.get("p2g_reduce", 0.0) for x in xfer_steps])
        t_g2p = med([x["g2p"] for x in xfer_steps])
        Nn = levels[0][0]
        tb = 43 * s * Np + 7 * s * Nn
        transfers = {"p2g_ms": t_p2g, "g2p_ms": t_g2p, "mparticles_per_s": Np / ((t_p2g + t_g2p) * 1e-3) / 1e6, "algorithmic_bytes": tb, "per_step_ms": [{k: round(v, 4) for k, v in x.items()} for x in xfer_steps], "statistic": "median over %d profiled steps" % nprof,
                     "achieved_GBps": tb / ((t_p2g + t_g2p) * 1e-3) / 1e9, "frac_of_hbm_peak": tb / ((t_p2g + t_g2p) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # what the two kernels actually move (recorded PMC counters, see pmc_traffic): G2P also carries the fused strain update (Fn in, F out:
        # 18 s per particle that SURVEY 8(d)'s transfer-only figure does not count) and is bound by THAT traffic; P2G is bound by its FP64 issue
        if args.config == "C2" and not args.cells:
            tp, src = pmc_traffic(SYMBOL["p2g"], "double" if s == 8 else "float")
            tg, _ = pmc_traffic(SYMBOL["g2p"], "double" if s == 8 else "float")
            if tp and tg:
                tr, _ = pmc_traffic("hot::k_tile_reduce<T", "double" if s == 8 else "float")  # (the launches of both scatters' reductions, averaged)
                transfers.update({"pmc_bytes_p2g_kernel": tp, "pmc_bytes_g2p_kernel": tg, "pmc_source": src,
                                  "g2p_traffic_GBps": tg / (t_g2p * 1e-3) / 1e9, "g2p_traffic_frac_of_hbm_peak": tg / (t_g2p * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  # what P2G (+ its tile reduction) and G2P MOVE over the time they take, against the 8 TB/s: the fused design's own figure
                                  # (SURVEY 8(d)'s bytes leave out the partial tiles and the strain update G2P carries)
                                  "frac_of_hbm_peak_traffic": (tp + (tr or 0.0) + tg) / ((t_p2g + t_g2p) * 1e-3) / 1e9 / HBM_PEAK_GBS})
        prof_top = sorted(((k, round(v["ms"] / nprof, 3), v["calls"] // nprof) for k, v in groups.items()), key=lambda x: -x[1])[:14]
        del pctx

    # ---- CPU baseline: the oracle (the reference's TBB decomposition restated with OpenMP) on the GPU box's host cores, timed on the
    # headline configuration itself — one whole time step per variant, same inputs, same solver knobs — in two variants
    # (SURVEY.md §8d): "faithful" leaves serial what the reference leaves serial, "fair" parallelises those sections too.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        import subprocess
        # the checker is rebuilt here, on the host that times it: -march=native must mean THIS machine's cores
        rebuilt = subprocess.call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"]) == 0
        from tests.oracle_lib import load_oracle
        ora = load_oracle()
        nc = args.cpu_cells or n1
        cap = dict(max_iterations=args.cpu_iters) if args.cpu_iters > 0 else {}
        sample = cloud if nc == n1 else parallel.shard_cloud(cfg, 0, 1, n=nc)

        def per_iter(st):
            return (st["ms_solve"] - st["ms_hessian"] - st["ms_mg_build"]) / max(st["iterations"], 1)
        res = {}
        for nm in ("faithful", "fair"):
            if nm == "fair":
                os.environ["HOT_ORACLE_FAIR"] = "1"  # read by the oracle when the context is created
            try:
                c = make_ctx(ora, sample, cfg, device=local, **cap)
                res[nm] = c.advance(dt)  # the first step of the run (the GPU's first step is timed beside it below)
                del c
            finally:
                os.environ.pop("HOT_ORACLE_FAIR", None)
        g = make_ctx(lib, sample, cfg, device=local, **cap)
        res["gpu"] = g.advance(dt)
        del g
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            model = "unknown"
        f, fa, gp = res["faithful"], res["fair"], res["gpu"]
        threads = int(ora.lib.hoto_num_threads())
        cpu = {"value": per_iter(f), "unit": "ms per L-BFGS iteration", "cores": threads, "kind": "port", "variant": "faithful",
               "cpu_model": model, "host_cores_visible": cores, "omp_threads": threads, "oracle_rebuilt_on_this_host": rebuilt,
               "sample": f"{args.config} itself: {nc}^3-cell cube, {sample['X'].shape[0]} particles, {f['num_nodes']} nodes, "
                         + (f"the first {f['iterations']} L-BFGS iterations of the first time step" if cap else f"the whole first time step ({f['iterations']} L-BFGS iterations)")
                         + ", same solver knobs, one run per variant, no warm-up; Hessian + hierarchy build timed separately (cpu_build_ms)",
               "fair_value": per_iter(fa), "fair_iterations": fa["iterations"], "iterations": f["iterations"],
               "gpu_same_step_ms_per_iter": per_iter(gp), "gpu_iterations": gp["iterations"],
               "speedup_per_iteration_vs_faithful": per_iter(f) / max(per_iter(gp), 1e-9), "speedup_per_iteration_vs_fair": per_iter(fa) / max(per_iter(gp), 1e-9),
               "cpu_step_ms": f["ms_total"], "cpu_fair_step_ms": fa["ms_total"], "gpu_step_ms": gp["ms_total"],
               "step_speedup_vs_faithful": f["ms_total"] / gp["ms_total"], "step_speedup_vs_fair": fa["ms_total"] / gp["ms_total"],
               "cpu_p2g_g2p_mparticles_per_s": sample["X"].shape[0] / ((f["ms_p2g"] + f["ms_g2p"]) * 1e-3) / 1e6,
               "cpu_build_ms": f["ms_hessian"] + f["ms_mg_build"], "cpu_fair_build_ms": fa["ms_hessian"] + fa["ms_mg_build"], "gpu_build_ms": gp["ms_hessian"] + gp["ms_mg_build"],
               "cpu_sort_ms": f["ms_sort"], "cpu_fair_sort_ms": fa["ms_sort"]}

    by_rank = None
    if world > 1:  # what every rank handed to the collectives and where its step went (a sharded run's balance, not only rank 0's view)
        import torch
        last = stats[-1]
        mine = torch.tensor([float(Np), float(last["comm_calls"]), float(last["comm_calls_index"]), float(last["comm_bytes_index"]), float(last["comm_bytes_data"]), last["ms_sort"], last["ms_p2g"], last["ms_begin"],
                             last["ms_hessian"], last["ms_mg_build"], last["ms_solve"], last["ms_g2p"], last["ms_total"]], dtype=torch.float64, device=torch.device("cuda", local) if args.backend == "nccl" else "cpu")
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        keys = ("particles", "collective_calls", "index_collective_calls", "index_bytes", "data_bytes", "ms_sort", "ms_p2g", "ms_begin", "ms_hessian", "ms_mg_build", "ms_solve", "ms_g2p", "ms_total")
        by_rank = [{k: (int(v) if i < 5 else round(float(v), 2)) for i, (k, v) in enumerate(zip(keys, t.tolist()))} for t in allv]
    if rank == 0:
        out = {
            "metric": "ms per nonlinear (L-BFGS) iteration; P2G+G2P Mparticles/s; achieved HBM GB/s vs roofline",
            "value": ms_per_iter, "unit": "ms/iter", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / max(args.steps, 1), "higher_is_better": False, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64" if s == 8 else "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: one {n}^3-cell cube x {cfg['ppc']} ppc ({'the whole body over all GPUs' if args.scaling == 'strong' and world > 1 else str(n1) + '^3 cells per GPU'}), fixed-corotated E={cfg['E']:g} nu={cfg['nu']}, dx=0.01, dt=1/24, "
                                   f"-lsolver 3 -mg_level {cfg['levelCnt']} -smoother 5 -coarseSolver 2 --project --linesearch --bcproject --usecn -cneps 1e-7"
                                   + ({1: f", von Mises return mapping (yield {cfg.get('yield_stress', 0):g})", 2: ", snow plasticity return mapping"}.get(cfg.get("plasticity", 0), "")),
                       "particles_per_gpu": int(total_particles / world), "particles_total": int(total_particles), "nodes_total": stats[-1]["num_nodes"], "levels": stats[-1]["num_levels"],
                       "parallelism": "1 GPU" if world == 1 else f"one connected body sharded over {world} ranks: particle ranges of the sort order, node tiles summed between the ranks sharing a block, "
                                                                 f"row-partitioned operators with halo gathers, partitioned vector algebra, coloured Gauss-Seidel {('rank-local (processor-block), one exchange per symmetric sweep' + (', l1-scaled diagonal' if args.shard_gs == 2 else '')) if args.shard_gs else 'colour-synchronous across ranks'} ({args.backend})"},
            "iterations_per_step": iters / max(args.steps, 1),
            "hessian_mg_build_ms_per_step": build_ms / max(args.steps, 1),
            "ms_per_iter_build_amortised": solve_ms / max(iters, 1),
            "p2g_g2p_mparticles_per_s": transfers["mparticles_per_s"] if transfers else None,
            "communicator": (None if comm is None else ("native RCCL on the context's stream" if isinstance(comm, str) and not used_fallback_comm else f"torch.distributed ({args.backend})")),
            "comm_calls_per_step": ({k: v / max(args.steps + args.warmup, 1) for k, v in comm.calls.items()} if (comm is not None and not isinstance(comm, str)) else None),
            "comm_per_step_rank0": ({"collective_calls": stats[-1]["comm_calls"], "index_bytes": stats[-1]["comm_bytes_index"], "data_bytes": stats[-1]["comm_bytes_data"]} if world > 1 else None),
            "last_step_by_rank": by_rank,
            "roofline": roof, "transfers": transfers, "cpu_baseline": cpu, "kernel_ms_per_step_top": prof_top,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
