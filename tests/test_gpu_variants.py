"""The library picks kernel variants by problem size (block GS: one launch per colour on large levels, one chained launch
per half sweep with point-to-point block flags on small ones; sub-block size 32 / 64); hot_config.gs_chain / gs_sub_block
override the choice.  The parity tests use small problems, so without this file only the small-problem variants would be
compared with the oracle.  First-generation kernels and launch-structure alternatives live only in the A/B build of the
library (libhotmi355x_ab.so, -DHOT_AB_KERNELS), where environment variables select them.  Each case re-runs the relevant
parity tests in a subprocess: HOT_TEST_CFG carries hot_config overrides that tests/pipeline_checks.make_ctx applies,
HOT_AMD_AB=1 makes hot_amd.load() pick the A/B build."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SOLVER = "tests/test_gpu_solver.py"
CASES = [
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32"}, SOLVER, "smoothers or vcycle or iterates"),
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=16"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=64"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_TEST_CFG": "gs_chain=2,gs_sub_block=32"}, SOLVER, "smoothers or vcycle or iterates"),
    ({"HOT_TEST_CFG": "gs_chain=2,gs_sub_block=16"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_GS_NO_WINV": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # chained whole-block passes with the 64-step substitution instead of the product with the blocks' inverse images (k_gs_winv)
    ({"HOT_TEST_CFG": "gs_chain=2", "HOT_GS_NO_WINV": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_GS_PASS_COUNTERS": "1"}, SOLVER, "smoothers or vcycle or iterates"),
    ({"HOT_TEST_CFG": "gs_chain=2,gs_sub_block=32", "HOT_GS_PASS_COUNTERS": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_TEST_CFG": "gs_chain=2,gs_sub_block=32", "HOT_GS_BLOCK_FLAGS": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # hand-off through per-block sweep stamps
    ({"HOT_GS_BLOCK_FLAGS": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_OFF_WAVES": "4100"}, SOLVER, "smoothers or vcycle"),  # a grid that is no multiple of 8: off-block steps dealt round robin instead of in per-XCD runs
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_OFF_WAVES": "64"}, SOLVER, "smoothers or vcycle"),  # few wavefronts: many steps per wavefront, odd and even step counts
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_V1": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # first-generation k_gs_block instead of the off-block / substitution pair
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_V1": "1", "HOT_GS_SPLIT_LAUNCHES": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_SIMPLE_GS": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_GS_FULL_RESIDUAL": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_MG_FULL_SPMV": "1"}, SOLVER, "vcycle or iterates"),
    ({"HOT_LBFGS_UNFUSED": "1"}, SOLVER, "iterates"),
    ({"HOT_CG_UNFUSED": "1"}, SOLVER, "smoothers or vcycle or iterates"),
    ({"HOT_CG_LAUNCHES": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # three launches per PCG iteration instead of the persistent launch on small top levels
    ({"HOT_GS_FAKE_TIMEOUT": "2"}, SOLVER, "smoothers or vcycle or iterates"),  # a chained sweep "times out" at the second synchronisation of every context: the operation is redone with one launch per pass
    ({"HOT_HESSIAN_V1": "1"}, SOLVER, "hessian_and_hierarchy"),
    ({"HOT_HESSIAN_TILES": "1"}, SOLVER, "hessian_and_hierarchy"),  # rounds 2 - 4: k_hessian_tiles2, particle chunks staged in LDS, pair phase with LDS atomics
    ({"HOT_HESSIAN_TILES_V1": "1"}, SOLVER, "hessian_and_hierarchy"),
    ({"HOT_HESSIAN_MFMA": "1"}, SOLVER, "hessian_and_hierarchy"),  # pair phase on v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32
    ({"HOT_P2G_V1": "1"}, "tests/test_gpu_transfer.py", "sort_p2g_g2p or transfer_properties"),
    ({"HOT_P2G_CELLS1": "1"}, "tests/test_gpu_transfer.py", "sort_p2g_g2p or transfer_properties"),
    ({"HOT_G2P_V1": "1"}, "tests/test_gpu_transfer.py", "sort_p2g_g2p or transfer_properties"),  # node-by-node sums instead of the sum-factorised ones
    ({"HOT_FORCE_V1": "1"}, "tests/test_gpu_force.py", "objective_pieces"),
    ({"HOT_FORCE_CELLS1": "1"}, "tests/test_gpu_force.py", "objective_pieces"),  # round 2's 3-node items with 27 staged scalars
]


@pytest.mark.gpu
@pytest.mark.parametrize("env,path,expr", CASES, ids=[" ".join(f"{k}={v}" for k, v in c[0].items()) for c in CASES])
def test_kernel_variant_parity(env, path, expr):
    e = dict(os.environ)
    e.update(env)
    if any(k != "HOT_TEST_CFG" for k in env):
        e["HOT_AMD_AB"] = "1"  # environment switches exist only in the A/B build
    r = subprocess.run([sys.executable, "-m", "pytest", path, "-x", "-q", "-m", "gpu", "-k", expr, "-p", "no:cacheprovider"], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout
