"""Iteration drift of the rank-local (processor-block) GS against the colour-synchronous one: PN + MG-PCG, 6 Newton steps, outer PCG
iterations (= V-cycles).  Ranks share the one GPU of the test box (gloo).  python tools/gs_drift.py <cells> <ranks> <partition_min_rows>"""
import sys
sys.path.insert(0, "/root/repo")
from tests import multirank_worker as mw


def main():
    n, world, minrows = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    kw = dict(lsolver=2, levelCnt=3, cneps=1e-8, max_iterations=6)
    out = []
    for mode in (0, 1):
        r = mw.launch(world, "hip", n, 1, dict(kw, shard_gs=mode), partition_min_rows=minrows, timeout=280)
        s = r[0]["stats"]
        out.append((s["vcycles"], s["linear_iterations"], r[0]["comm_calls"]["allgather"]))
    print("n", n, "ranks", world, "V-cycles / top-level PCG iterations / all-gathers: colour-synchronous", out[0], "rank-local", out[1], flush=True)


if __name__ == "__main__":
    main()
