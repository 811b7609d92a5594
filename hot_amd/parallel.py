"""Workload helpers of the benchmark and the tests for one process per GPU.

The multi-GPU decomposition itself lives in the library (hot_amd/csrc/shard.hip, hot_set_comm) and in hot_amd/dist.py
(the collectives over torch.distributed, the split of a cloud into page-order shards): ONE connected body is sharded over
the ranks in "halo mode" — particle runs of the global page order per rank, node tiles summed pairwise between the ranks
that share a block, matrix rows owned by one rank each, DOF vectors on owned rows plus halos, batched scalar all-reduces,
colour-synchronous or rank-local Gauss-Seidel.  This module only builds the synthetic
body of a BASELINE configuration, hands a rank its shard, and provides the bench clock (max over ranks).

Weak scaling: the body grows with the number of ranks so that every GPU keeps the single-GPU configuration's particle
count (`cells_for_world`); for world == 1 it is exactly the configuration of hot_amd/synth.CONFIGS."""
import numpy as np

from . import synth


def cells_for_world(n, world):
    """Cube edge (cells) of the one body that gives `world` ranks the particle count an n^3 body gives one rank."""
    return int(round(n * world ** (1.0 / 3.0)))


def body_cloud(cfg, n=None, dx=0.01):
    """The whole synthetic body of configuration `cfg` (hot_amd/synth.CONFIGS entry) with an n^3-cell cube."""
    n = n or cfg["n"]
    c = synth.cube_cloud(n, ppc=cfg["ppc"], dtype=cfg["dtype"], E=cfg["E"], nu=cfg["nu"], rho=cfg["rho"], corner=(5.0, 5.0, 5.0), seed=123, dx=dx)
    c["corner"] = (5.0, 5.0, 5.0)
    c["cells"] = n
    return c


def shard_cloud(cfg, rank, world, n=None, dx=0.01):
    """Rank `rank`'s particles of the one body `world` ranks share: the whole body for world == 1, otherwise its contiguous
    range of the SPGrid page order (hot_amd/dist.shard_by_page_order).  `n` = cube edge of the WHOLE body."""
    c = body_cloud(cfg, n, dx)
    if world == 1:
        return c
    from . import dist as hdist
    s = hdist.shard_by_page_order(c, rank, world)
    s["corner"], s["cells"] = c["corner"], c["cells"]
    return s


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
