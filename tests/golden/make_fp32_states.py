"""Start states of tests/test_gpu_solver.py::test_three_time_steps_fp32_against_fp32_oracle: the particle data (X, V, C, F; float32) of the 8^3-cell test
cube after one .. five time steps of dt = 1/24 taken by the HIP library's converged fp32 solve (cneps = 1e-4).  The fp32 trajectory of this
problem is chaotic (level-0 system of cond ~ 1 / eps_float): a last-bit change anywhere in the device code moves the states, and on some of them the
oracle's own float solve breaks down, so the test compares bounded numbers of iterations from THESE fixed states instead of from whatever the build
under test produces.  Run on a GPU box:  python tests/golden/make_fp32_states.py  -> gpurun_out/fp32_states.npz (copy to tests/golden/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hot_amd
from hot_amd import synth

lib = hot_amd.load()
c = synth.cube_cloud(8, ppc=8, dtype=np.float32)
o, nrm = synth.sticky_floor(5.0, c["dx"])
state = dict(X=c["X"], V=c["V"], C_=None, F=None)
out = {}
for step in (1, 2, 3, 4, 5):
    ctx = lib.context(dtype=0, dx=c["dx"], gravity=(0, -9.8, 0), levelCnt=2, max_iterations=300, cneps=1e-4)
    ctx.set_particles(state["X"], state["V"], c["mass"], c["vol"], c["mu"], c["lam"], C_=state["C_"], F=state["F"])
    ctx.set_sticky_halfspaces(o, nrm)
    st = ctx.advance(1.0 / 24)
    assert st["converged"] == 1
    p = ctx.get_particles()
    state = dict(X=p["X"], V=p["V"], C_=p["C"], F=p["F"])
    for k, v in (("X", p["X"]), ("V", p["V"]), ("C", p["C"]), ("F", p["F"])):
        out["%s%d" % (k, step)] = np.asarray(v, np.float32)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/fp32_states.npz", **out)
print("wrote gpurun_out/fp32_states.npz", {k: v.shape for k, v in out.items()})
# The committed tests/golden/fp32_states.npz holds two of these, renumbered 1 and 2: the states after ONE and after FOUR steps.  On the state after two
# steps (a node of mass 5e-13 whose coarse-level row has eigenvalues of 7e-14) the oracle's float PCG on the top level divides by a vanished du'A du and
# every trial energy is NaN — the reference's cg_smooth has no guard either —, so nothing can be compared from it; checked on CPU with the oracle:
#   python - <<'PY'
#   ... for every state: hoto solve with max_iterations 1 and 4, wide sums, isfinite(dv) ...   (states 1, 3, 4, 5 finite; 2 not)
#   PY
