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
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_PAIR": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # round 4 / 5: the colour pass as the kernel pair k_gs_offblock + k_gs_subst (what a row-partitioned level runs) instead of the one launch k_gs_colour
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_PAIR": "1", "HOT_GS_OFF_WAVES": "64"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_SUBST_D": "4"}, SOLVER, "smoothers or vcycle"),  # image columns in flight per substitution wavefront
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_SUBST_D": "6"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_SUBST_D": "8"}, SOLVER, "smoothers or vcycle or iterates"),
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_NO_TURN": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # the backward sweep's first colour as a launch of its own instead of riding on the forward sweep's last (k_gs_colour<.., TURN>)
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_OFF_WAVES": "4100"}, SOLVER, "smoothers or vcycle"),  # a grid that is no multiple of 8: off-block steps dealt round robin instead of in per-XCD runs
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_OFF_WAVES": "64"}, SOLVER, "smoothers or vcycle"),  # few wavefronts: many steps per wavefront, odd and even step counts
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_V1": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # first-generation k_gs_block instead of the off-block / substitution pair
    ({"HOT_TEST_CFG": "gs_chain=1,gs_sub_block=32", "HOT_GS_V1": "1", "HOT_GS_SPLIT_LAUNCHES": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_SIMPLE_GS": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_GS_FULL_RESIDUAL": "1"}, SOLVER, "smoothers or vcycle"),
    ({"HOT_MG_FULL_SPMV": "1"}, SOLVER, "vcycle or iterates"),
    ({"HOT_LBFGS_UNFUSED": "1"}, SOLVER, "iterates"),
    ({"HOT_LS_NO_BATCH": "1"}, SOLVER, "iterates or objective_concept or knobs"),  # line-search trials one pass each (their own gradient gather) instead of batches of 2 / 4 / 8 on F(alpha) = A0 + alpha A1 (trial_batch, force.hip)
    ({"HOT_CG_UNFUSED": "1"}, SOLVER, "smoothers or vcycle or iterates"),
    ({"HOT_CG_STREAM": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # the persistent top-level PCG reading its rows from memory every iteration instead of holding them in registers (k_cg_persist<T, 2>)
    ({"HOT_CG_WGS": "40"}, SOLVER, "smoothers or vcycle or iterates"),  # fewer workgroups than rows / 32: the streaming version on a small level
    ({"HOT_CG_LAUNCHES": "1"}, SOLVER, "smoothers or vcycle or iterates"),  # three launches per PCG iteration instead of the persistent launch on small top levels
    ({"HOT_GS_FAKE_TIMEOUT": "2"}, SOLVER, "smoothers or vcycle or iterates"),  # a chained sweep "times out" at the second synchronisation of every context: the operation is redone with one launch per pass
    ({"HOT_HESSIAN_V1": "1"}, SOLVER, "hessian_and_hierarchy"),
    ({"HOT_HESSIAN_TILES": "1"}, SOLVER, "hessian_and_hierarchy"),  # rounds 2 - 4: k_hessian_tiles2, particle chunks staged in LDS, pair phase with LDS atomics
    ({"HOT_HESSIAN_TILES_V1": "1"}, SOLVER, "hessian_and_hierarchy"),
    ({"HOT_HESSIAN_MFMA": "1"}, SOLVER, "hessian_and_hierarchy"),  # pair phase on v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32
    ({"HOT_P2G_CELLS2": "1"}, "tests/test_gpu_transfer.py", "sort_p2g_g2p or transfer_properties"),  # rounds 2 - 5: one workgroup per particle group with register staging (production since round 6: k_p2g_stream, persistent workgroups fed by LDS-DMA)
    ({"HOT_P2G_WGS_PER_CU": "1"}, "tests/test_gpu_transfer.py", "sort_p2g_g2p or transfer_properties"),  # k_p2g_stream with half the workgroups: twice the units per workgroup
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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [1, 0])
def test_batched_line_search_trials_are_the_single_ones(dtype):
    """A search that has to halve 3 .. 12 times, from the same state, with the trial energies taken from batches (one pass for 2 / 4 / 8 of them,
    Ctx::trial_batch) and one pass each (A/B switch HOT_LS_NO_BATCH): the same step lengths (hence trial counts), the same accepted point
    and residual (the batch evaluates F(alpha) = A0 + alpha A1, the single pass gathers the trial point's own gradient: equal to rounding)."""
    import numpy as np
    import hot_amd
    from tests import pipeline_checks as pc
    lib = hot_amd.HotLib(hot_amd.AB_LIB_PATH)
    try:
        d0 = None
        for scale in (8.0, 100.0, 4000.0):
            out = []
            for no_batch in (False, True):
                os.environ.pop("HOT_LS_NO_BATCH", None)
                if no_batch:
                    os.environ["HOT_LS_NO_BATCH"] = "1"
                ctx, c = pc.make_ctx(lib, n=8, dtype=dtype, levelCnt=2, ls_energy_only=2)
                pc.prepare(ctx)
                ctx.update_state(ctx.get_dv())
                r = ctx.residual()
                if d0 is None:  # one direction for every context (two assemblies of the Hessian differ in the last bits: the order of the LDS atomics)
                    ctx.build_hessian(), ctx.build_mg()
                    d0 = ctx.project(ctx.vcycle(r))
                d = d0 * scale  # an overlong step: the energy rises until it has been halved often enough
                dd, r2, alpha = ctx.line_search(d, 1.0)
                out.append((alpha, None, dd, r2, ctx.get_dv()))
                ctx.line_search(d, 1.0)  # (a second search, whose first batch is sized by the first search's count; from the accepted point there is nothing to gain along d, where it stops is round-off)
            a, b = out
            assert a[0] == b[0], (scale, a[0], b[0])
            assert a[0] <= 0.25, (scale, a[0])  # (three trials or more)
            for k in (2, 3, 4):  # (the scatters' LDS-atomic order differs from context to context: equal to round-off, bitwise where no scatter is involved)
                err = np.abs(a[k].astype(np.float64) - b[k]).max() / max(np.abs(b[k]).max(), 1e-300)
                assert err < (1e-11 if dtype == 1 else 1e-4), (scale, k, err)
    finally:
        os.environ.pop("HOT_LS_NO_BATCH", None)
