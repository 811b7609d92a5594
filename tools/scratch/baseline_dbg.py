import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import hot_amd, pipeline_checks as pc
from hot_amd.binding import HALFSPACE, SLIP, STICKY
lib = hot_amd.load()
objs = [dict(shape=HALFSPACE, type=STICKY, p0=(0, 5.0 + 0.0049, 0), p1=(0, 1.0, 0)), dict(shape=HALFSPACE, type=SLIP, p0=(5.0 + 0.0151, 0, 0), p1=(0.8, 0, 0.6))]
ctx, c = pc.make_ctx(lib, n=12, dtype=1, bc=False, levelCnt=3, cneps=1e-7, useBaselineMultigrid=1, boundaryType=1, max_iterations=60)
ctx.set_collision_objects(objs)
pc.prepare(ctx)
ctx.update_state(ctx.get_dv())
ctx.build_hessian()
print("build_mg", flush=True)
ctx.build_mg()
print("ok")
