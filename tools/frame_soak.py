"""Soak run of the frame driver (not a test): fast bodies, CFL-limited substeps, several frames.  python tools/frame_soak.py [C1|C2|...] [cells]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    cfg = dict(synth.CONFIGS[name])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["n"]
    cloud = parallel.shard_cloud(cfg, 0, 1, n=n)
    cloud["V"] = (cloud["V"] * 20).astype(cloud["V"].dtype)  # fast enough for the CFL limit to bite
    ctx = bench.make_ctx(hot_amd.load(), cloud, cfg, cfl=0.6)
    for f in range(8):
        nsub, its, st = ctx.advance_frame(1.0 / 24)
        p = ctx.get_particles()
        print("frame", f, "substeps", nsub, "iterations", its, "converged", st["converged"], "|V| max %.3g" % np.abs(p["V"]).max(), "finite", bool(np.isfinite(p["X"]).all()), flush=True)


if __name__ == "__main__":
    main()
