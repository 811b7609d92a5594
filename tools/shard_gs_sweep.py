"""Iteration counts of one converged time step of a <cells>^3-cell cube (8 particles per cell, fp64, 3 levels, the bench's solver line) over <ranks> ranks
sharing the box's one GPU through gloo, for the three ways the coloured GS crosses ranks: colour-synchronous (hot_config.shard_gs = 0: the single-rank
iterates), rank-local (1), rank-local with the l1-scaled diagonal (2).   python tools/shard_gs_sweep.py <cells> <ranks> [<cells> <ranks> ...]"""
import sys
sys.path.insert(0, "/root/repo")
from tests import multirank_worker as mw


def main():
    args = [int(a) for a in sys.argv[1:]]
    kw = dict(lsolver=3, levelCnt=3, smoother=5, coarseSolver=2, project=1, linesearch=1, systemBCProject=1, useCN=1, cneps=1e-7, max_iterations=400)
    for n, world in zip(args[::2], args[1::2]):
        out = {}
        for gs in (0, 2, 1):
            try:
                r = mw.launch(world, "hip", n, 1, dict(kw, shard_gs=gs), steps=1, partition_min_rows=4096, timeout=900)
                out[gs] = (r[0]["stats"]["iterations"], r[0]["stats"]["converged"])
            except Exception as e:  # (a run that does not finish in time)
                out[gs] = ("no result: " + type(e).__name__, 0)
        base = out[0][0]
        print("cells %d^3 over %d ranks: colour-synchronous %s iterations; l1-scaled rank-local %s (%s); plain rank-local %s (%s)" % (
            n, world, base, out[2][0], "%+.0f %%" % (100.0 * (out[2][0] - base) / base) if isinstance(out[2][0], int) and out[2][1] else "not converged",
            out[1][0], "%+.0f %%" % (100.0 * (out[1][0] - base) / base) if isinstance(out[1][0], int) and out[1][1] else "not converged"), flush=True)


if __name__ == "__main__":
    main()
