import sys
sys.path.insert(0, "/root/repo")
from tests import multirank_worker as mw
import hot_amd
n = int(sys.argv[1]); world = int(sys.argv[2])
kw = dict(lsolver=3, levelCnt=3, cneps=1e-7, max_iterations=400)
ref = mw.single(hot_amd.load(), n, 1, kw)
r = mw.launch(world, "hip", n, 1, dict(kw, shard_gs=1), partition_min_rows=1)
print("n", n, "world", world, "single", ref["stats"]["iterations"], "rank-local", r[0]["stats"]["iterations"], "rel dv", mw.rel(r[0]["dv"], ref["dv"]), "allgathers", r[0]["comm_calls"]["allgather"], flush=True)
