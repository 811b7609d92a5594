#!/bin/bash
# Round-4 experiment batch (one gpurun call): parity of the touched kernels, then A/B timings.  Output under gpurun_out/exp_r4b/.
O=gpurun_out/exp_r4b; mkdir -p $O
export HOT_PROF_TOP=14
timeout 900 python -m pytest tests/test_gpu_transfer.py tests/test_gpu_golden.py tests/test_gpu_solver.py tests/test_gpu_fullsize.py -q -m gpu -x -k "not convergence and not C3 and not C4 and not C5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "G2P or GS_V1 or SPLIT" > $O/pytest_var.log 2>&1; echo "variants rc=$?"; tail -3 $O/pytest_var.log
echo "== transfers, new G2P"; HOT_COLD=1 HOT_PRESTEPS=2 timeout 300 python tools/p2g_time.py C2 C3 2>&1 | grep -v amdgpu | tee $O/p2g_new.log
echo "== transfers, node-by-node G2P"; HOT_AMD_AB=1 HOT_G2P_V1=1 HOT_COLD=1 HOT_PRESTEPS=2 timeout 300 python tools/p2g_time.py C2 C3 2>&1 | grep -v amdgpu | tee $O/p2g_old.log
echo "== C2 step, default"; timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee $O/prof_default.log
echo "== C2 step, off-block steps dealt round robin (old)"; HOT_AMD_AB=1 HOT_GS_OFF_WAVES=4100 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee $O/prof_rr.log
echo "== C2 step, kernel pair on level 1 too"; HOT_SOAK_CFG=gs_sub_block=32 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee $O/prof_pairL1.log
