import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd
from tests import multirank_worker as mw
kw = dict(lsolver=3, levelCnt=3, max_iterations=5, cneps=1e-7)

def main():
    ref = mw.single(hot_amd.load(), 8, 1, kw)
    for rep in (1, 0):
        ranks = mw.launch(2, "hip", 8, 1, dict(kw, shard_replicated=rep), partition_min_rows=1)
        for r, o in enumerate(ranks):
            a, b = o["id2coord"], ref["id2coord"]
            same = a.shape == b.shape and np.array_equal(a, b)
            print("replicated" if rep else "halo", "rank", r, "nodes", a.shape, b.shape, "equal", same, "particles", len(o["ids"]))
            if not same and a.shape == b.shape:
                bad = np.where((a != b).any(1))[0]
                print("  first differing ids", bad[:10], "count", len(bad), "same set:", set(map(tuple, a)) == set(map(tuple, b)))
                print("  a", a[bad[:3]], "b", b[bad[:3]])


if __name__ == "__main__":
    main()
