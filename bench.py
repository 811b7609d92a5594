#!/usr/bin/env python3
"""bench.py — HOT hot path on MI355X: one "step" = one backward-Euler time step (sort -> P2G -> L-BFGS/Galerkin-MG
solve -> G2P) of the BASELINE.json config-2 stand-in (63^3-cell cube, 8 ppc = 2.0 M particles,
fp64, 3 MG levels, -lsolver 3 -smoother 5 -coarseSolver 2 --project --linesearch --bcproject --usecn).

Prints ONE JSON line on rank 0.  `value` = ms per nonlinear (L-BFGS) iteration, Hessian + hierarchy build excluded
(reported separately and amortised); extra keys carry the P2G+G2P Mparticles/s half of the metric.
`roofline` = the kernel with the largest share of the timed region, from HIP events recorded on the library's
launch stream (hot_config.profile); `cpu_baseline` = the CPU oracle (a port of the reference's TBB decomposition to
OpenMP) on a bounded sample, with the GPU timed on that same sample next to it.

N > 1: one process per GPU (torch.distributed / RCCL for the barriers and the max-over-ranks clock); every rank
advances its own spatial shard of the same size (weak scaling).  The shards are not yet coupled by halo exchange
 — `config.parallelism` says so.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def host_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def make_ctx(lib, cloud, cfg, device=0, profile=0, **over):
    from hot_amd import synth
    kw = dict(dtype=1 if cloud["X"].dtype == np.float64 else 0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=cfg["levelCnt"], device=device, profile=profile)
    kw.update(over)
    ctx = lib.context(**kw)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    o, n = synth.sticky_floor(cloud["corner"][1], cloud["dx"])
    ctx.set_sticky_halfspaces(o, n)
    return ctx


def algorithmic_bytes(name, s, Np, levels):
    """SURVEY.md §8(d) per-launch algorithmic bytes of the named kernel (None if not modelled)."""
    base, _, lv = name.rpartition("_L")
    if base in ("spmv", "gs_forward", "gs_backward") and lv.isdigit():
        N, nnzb = levels[int(lv)]
        if base == "spmv":
            return nnzb * (9 * s + 4) + N * 6 * s
        off = max(nnzb - N, 0) / 2.0  # strictly lower (or upper) blocks
        per_half_sweep = off * (9 * s + 4) + N * ((18 if base == "gs_forward" else 9) * s + 6 * s)
        return per_half_sweep / 8.0  # one launch per colour
    Nn = levels[0][0]
    if name == "p2g":
        return Np * 16 * s + Nn * 4 * s
    if name == "g2p":
        return Nn * 3 * s + Np * (3 * s + 24 * s) + Np * 18 * s  # + F in/out (evolveStrain is fused into the kernel)
    if name == "state_update_force":
        return Np * 24 * s + Nn * 6 * s
    if name == "hessian_assemble":
        return Np * 24 * s + levels[0][1] * 9 * s
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--cells", type=int, default=0, help="override the cube edge (cells) for quick runs")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-cells", type=int, default=24)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cores = host_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(cores, 32))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    import torch
    import hot_amd
    from hot_amd import synth

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP library has no CPU fallback)"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lib = hot_amd.load()
    cfg = dict(synth.CONFIGS[args.config])
    n = args.cells or cfg["n"]
    # weak scaling: rank r owns the block of cells shifted by r*(n+8) cells in x (disjoint sub-domains)
    corner = (5.0 + rank * (n + 8) * 0.01, 5.0, 5.0)
    cloud = synth.cube_cloud(n, ppc=cfg["ppc"], dtype=cfg["dtype"], E=cfg["E"], nu=cfg["nu"], rho=cfg["rho"], corner=corner, seed=123 + rank)
    cloud["corner"] = corner
    Np = cloud["X"].shape[0]
    s = 8 if cfg["dtype"] == np.float64 else 4
    dt = cfg["dt"]

    ctx = make_ctx(lib, cloud, cfg, device=local)
    for _ in range(args.warmup):
        ctx.advance(dt)
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(ctx.advance(dt))
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    iters = sum(st["iterations"] for st in stats)
    build_ms = sum(st["ms_hessian"] + st["ms_mg_build"] for st in stats)
    solve_ms = sum(st["ms_solve"] for st in stats)
    ms_per_iter = (solve_ms - build_ms) / max(iters, 1)
    if dist is not None:
        t = torch.tensor([ms_per_iter], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_per_iter = float(t.item())

    # ---- profiled pass (HIP events on the launch stream) for the roofline and the transfer half of the metric
    roof, transfers, prof_table = None, None, {}
    if rank == 0:
        pctx = make_ctx(lib, cloud, cfg, device=local, profile=1)
        pctx.advance(dt)
        pctx.profile_reset()
        nprof = max(1, min(2, args.steps))
        pst = [pctx.advance(dt) for _ in range(nprof)]
        prof_table = pctx.profile()
        levels = []
        for l in range(pst[-1]["num_levels"]):
            levels.append((pctx.level(l, coords=False)["nrows"], pctx.level_nnzb(l)))
        total_ms = sum(v["total_ms"] for v in prof_table.values())
        name = max(prof_table, key=lambda k: prof_table[k]["total_ms"])
        # group the per-colour GS launches / levels under the dominant name as recorded
        rec = prof_table[name]
        avg_ms = rec["total_ms"] / rec["calls"]
        ab = algorithmic_bytes(name, s, Np, levels)
        achieved = (ab / (avg_ms * 1e-3)) / 1e9 if ab else None
        roof = {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": None, "avg_launch_ms": avg_ms, "launches": rec["calls"], "algorithmic_bytes_per_launch": ab, "share_of_kernel_time": rec["total_ms"] / total_ms}
        tp = prof_table.get("p2g", {"total_ms": 0, "calls": 1})
        tg = prof_table.get("g2p", {"total_ms": 0, "calls": 1})
        t_p2g, t_g2p = tp["total_ms"] / tp["calls"], tg["total_ms"] / tg["calls"]
        Nn = levels[0][0]
        tb = 43 * s * Np + 7 * s * Nn
        transfers = {"p2g_ms": t_p2g, "g2p_ms": t_g2p, "mparticles_per_s": Np / ((t_p2g + t_g2p) * 1e-3) / 1e6, "algorithmic_bytes": tb,
                     "achieved_GBps": tb / ((t_p2g + t_g2p) * 1e-3) / 1e9, "frac_of_hbm_peak": tb / ((t_p2g + t_g2p) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        prof_top = sorted(((k, v["total_ms"] / nprof, v["calls"] // nprof) for k, v in prof_table.items()), key=lambda x: -x[1])[:12]
        del pctx

    # ---- CPU baseline: the oracle on a bounded sample, and the GPU on that same sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from tests.oracle_lib import load_oracle
        ora = load_oracle()
        nc = args.cpu_cells
        sample = synth.cube_cloud(nc, ppc=cfg["ppc"], dtype=cfg["dtype"], E=cfg["E"], nu=cfg["nu"], rho=cfg["rho"])
        sample["corner"] = (5.0, 5.0, 5.0)
        res = {}
        for nm, L in (("cpu", ora), ("gpu", lib)):
            c = make_ctx(L, sample, cfg, device=local)
            c.advance(dt)  # warm-up step (first-touch, thread pool)
            st = c.advance(dt)
            res[nm] = st
            del c
        def per_iter(st):
            return (st["ms_solve"] - st["ms_hessian"] - st["ms_mg_build"]) / max(st["iterations"], 1)
        cpu = {"value": per_iter(res["cpu"]), "unit": "ms per L-BFGS iteration", "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
               "sample": f"{nc}^3-cell cube, {sample['X'].shape[0]} particles, {res['cpu']['num_nodes']} nodes, 1 timed step of {res['cpu']['iterations']} iterations (same solver knobs)",
               "gpu_same_sample_ms_per_iter": per_iter(res["gpu"]), "speedup_same_sample": per_iter(res["cpu"]) / max(per_iter(res["gpu"]), 1e-9),
               "cpu_step_ms": res["cpu"]["ms_total"], "gpu_step_ms": res["gpu"]["ms_total"],
               "cpu_p2g_g2p_mparticles_per_s": sample["X"].shape[0] / ((res["cpu"]["ms_p2g"] + res["cpu"]["ms_g2p"]) * 1e-3) / 1e6,
               "cpu_build_ms": res["cpu"]["ms_hessian"] + res["cpu"]["ms_mg_build"], "gpu_build_ms": res["gpu"]["ms_hessian"] + res["gpu"]["ms_mg_build"]}

    if rank == 0:
        out = {
            "metric": "ms per nonlinear (L-BFGS) iteration; P2G+G2P Mparticles/s; achieved HBM GB/s vs roofline",
            "value": ms_per_iter, "unit": "ms/iter", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / max(args.steps, 1), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if s == 8 else "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n}^3-cell cube x {cfg['ppc']} ppc per GPU, fixed-corotated E={cfg['E']:g} nu={cfg['nu']}, dx=0.01, dt=1/24, "
                                   f"-lsolver 3 -mg_level {cfg['levelCnt']} -smoother 5 -coarseSolver 2 --project --linesearch --bcproject --usecn -cneps 1e-7",
                       "particles_per_gpu": Np, "nodes_per_gpu": stats[-1]["num_nodes"], "levels": stats[-1]["num_levels"],
                       "parallelism": "1 GPU" if world == 1 else f"{world} independent spatial shards (one per GPU, no halo coupling yet)"},
            "iterations_per_step": iters / max(args.steps, 1),
            "hessian_mg_build_ms_per_step": build_ms / max(args.steps, 1),
            "ms_per_iter_build_amortised": solve_ms / max(iters, 1),
            "p2g_g2p_mparticles_per_s": (transfers["mparticles_per_s"] * world) if transfers else None,
            "roofline": roof, "transfers": transfers, "cpu_baseline": cpu,
            "kernel_ms_per_step_top": prof_top if prof_table else None,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
