"""One-process-per-GPU sharding of the hot path (host side).

Round-1 decomposition: the scene is partitioned BY BODY.  Bodies whose particle clouds are separated by more than
the kernel support (2 cells) never share a grid node, so each rank owns whole bodies and the per-rank solves are
exactly the global solve restricted to those bodies — no halo exchange and no collective on the data path
(SURVEY.md §8e "Does the path shard naturally?"; a single connected body split across ranks needs the block-halo
exchange and is listed as next work in DESIGN.md §7).  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is
used only for the barrier / max-over-ranks clock and for gathering per-rank statistics.
"""
import numpy as np

from . import synth

GAP_CELLS = 8  # empty cells between neighbouring bodies: > kernel support (2) + motion margin


def body_corner(body, n, dx=0.01, origin=(5.0, 5.0, 5.0)):
    """Lower corner of body number `body` in the row of bodies along x."""
    return (origin[0] + body * (n + GAP_CELLS) * dx, origin[1], origin[2])


def assign_bodies(num_bodies, world):
    """Contiguous, balanced body ranges: rank r owns bodies [lo, hi)."""
    base, extra = divmod(num_bodies, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def shard_cloud(cfg, rank, world, n=None, bodies_per_rank=1, dx=0.01):
    """Particles owned by `rank`: its bodies' clouds concatenated.  Deterministic in (body index) only, so the union
    over ranks is independent of `world`."""
    n = n or cfg["n"]
    lo, hi = assign_bodies(world * bodies_per_rank, world)[rank]
    parts = []
    for b in range(lo, hi):
        c = synth.cube_cloud(n, ppc=cfg["ppc"], dtype=cfg["dtype"], E=cfg["E"], nu=cfg["nu"], rho=cfg["rho"], corner=body_corner(b, n, dx), seed=123 + b, dx=dx)
        parts.append(c)
    out = {k: np.concatenate([p[k] for p in parts]) for k in ("X", "V", "mass", "vol", "mu", "lam")}
    out["dx"] = dx
    out["corner"] = body_corner(lo, n, dx)
    out["bodies"] = (lo, hi)
    return out


def max_over_ranks(value, dist=None, device=None):
    """The bench clock: max of a host scalar over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
