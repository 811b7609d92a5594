"""What a rank hands to the collectives during ONE whole time step of a sharded body, by kind (hot_amd/csrc profile records commMB_*):
python tools/comm_breakdown.py [cells per rank edge] [ranks] [shard_gs] [partition_min_rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import multirank_worker as mw


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    gs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    minrows = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    n = round(per * world ** (1.0 / 3.0))
    for rep in ((0,) if os.environ.get("HOT_HALO_ONLY") else (0, 1)):
        kw = dict(lsolver=3, levelCnt=3, cneps=1e-7, shard_gs=gs, shard_replicated=rep, profile=1, shard_owner=int(os.environ.get("HOT_SHARD_OWNER", "0")))
        r = mw.launch(world, "hip", n, 1, kw, steps=1, partition_min_rows=minrows, timeout=1800)
        for rk in range(world):
            st = r[rk]["stats"]
            print("%d^3 cells over %d ranks, %s, shard_gs %d, min rows %d: %d iterations, rank %d: %d collective calls, data %.1f MB, index %.2f MB per step"
                  % (n, world, "replicated vectors" if rep else "halo mode", gs, minrows, st["iterations"], rk, st["comm_calls"], st["comm_bytes_data"] / 1e6, st["comm_bytes_index"] / 1e6))
            for k, v in sorted(r[rk]["profile"].items()):
                if k.startswith("commMB_") and v["total_ms"] >= 0.005:
                    print("    %-28s %6d calls %10.2f MB" % (k[7:], v["calls"], v["total_ms"]))


if __name__ == "__main__":
    main()
