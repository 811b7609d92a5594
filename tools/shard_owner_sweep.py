"""Row-ownership rule of a sharded run (hot_config.shard_owner) against (i) the balance of what the ranks hand to the collectives and (ii) the
L-BFGS iteration drift of the rank-local Gauss-Seidel sweep (hot_config.shard_gs = 1) against the colour-synchronous one (= the single-rank
iterates).  One whole converged time step of a cube of <cells>^3 cells (8 particles per cell, fp64, 3 levels, the bench's solver line) over <ranks>
ranks sharing the test box's one GPU through gloo.  python tools/shard_owner_sweep.py <cells> <ranks> [owner rules, default 1 2: first touch, page range]"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import multirank_worker as mw


def main():
    n, world = int(sys.argv[1]), int(sys.argv[2])
    rules = [int(a) for a in sys.argv[3:]] or [1, 2]
    kw = dict(lsolver=3, levelCnt=3, smoother=5, coarseSolver=2, project=1, linesearch=1, systemBCProject=1, useCN=1, cneps=1e-7, max_iterations=400)
    base = None
    for rule in rules:
        row = {}
        for gs in (0, 1):
            if gs == 0 and base is not None:
                row[0] = base  # colour-synchronous iterates do not depend on who owns a row
                continue
            r = mw.launch(world, "hip", n, 1, dict(kw, shard_gs=gs, shard_owner=rule), steps=1, partition_min_rows=4096, timeout=1200)
            its = r[0]["stats"]["iterations"]
            data = np.array([q["stats"]["comm_bytes_data"] for q in r], dtype=np.float64)
            row[gs] = (its, data)
            if gs == 0:
                base = row[0]
        (i0, d0), (i1, d1) = row[0], row[1]
        print("cells %d ranks %d shard_owner %d: iterations colour-synchronous %d, rank-local %d (drift %+.1f %%); data bytes per rank (rank-local run) min %.1f MB max %.1f MB ratio %.2f  %s"
              % (n, world, rule, i0, i1, 100.0 * (i1 - i0) / i0, d1.min() / 1e6, d1.max() / 1e6, d1.max() / d1.min(), np.round(d1 / 1e6, 1).tolist()), flush=True)


if __name__ == "__main__":
    main()
