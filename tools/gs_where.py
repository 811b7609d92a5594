"""Where does a finest-level coloured-GS launch spend its time?  TIMING experiment in the A/B build (wrong results): the same V-cycles
with parts of k_gs_block switched off (HOT_GS_DBG bits: 1 no substitution phase, 2 no x gathers, 4 no matrix value loads, 8 no phase A)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import hot_amd
from hot_amd import synth, parallel
lib = hot_amd.HotLib(hot_amd.AB_LIB_PATH)
cfg = synth.CONFIGS[sys.argv[1]]
cloud = parallel.shard_cloud(cfg, 0, 1, n=int(os.environ.get("GS_CELLS", cfg["n"])))
ctx = lib.context(dtype=1 if cfg["dtype"] == np.float64 else 0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=cfg["levelCnt"], profile=1)
ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
o, nrm = synth.sticky_floor(cloud["corner"][1], cloud["dx"])
ctx.set_sticky_halfspaces(o, nrm)
ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
ctx.update_state(ctx.get_dv())
ctx.build_hessian(), ctx.build_mg()
x = ctx.project(np.random.default_rng(1).standard_normal((ctx.Nn, 3)))
ctx.vcycle(x)
ctx.profile_reset()
for _ in range(6):
    ctx.vcycle(x)
t = ctx.profile()
L0 = ctx.level(0, coords=False)["nrows"]
print(json.dumps(dict({k: v["total_ms"] / v["calls"] for k, v in t.items() if k.startswith("gs_")}, nodes=L0 * 1e-3)))
''' % ROOT
for cname in sys.argv[1:] or ["C2"]:
    for flags in ([int(f) for f in os.environ["GS_FLAGS"].split(",")] if os.environ.get("GS_FLAGS") else (0, 1, 2, 4, 6, 7, 8, 9)):
        e = dict(os.environ, HOT_GS_DBG=str(flags))
        r = subprocess.run([sys.executable, "-c", CHILD, cname], env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {}
        print(cname, "flags", flags, " ".join("%s=%.1fus" % (k, 1e3 * v) for k, v in sorted(d.items()) if "symsweeps" not in k), flush=True)
