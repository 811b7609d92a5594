#!/bin/bash
O=gpurun_out/exp_r4d; mkdir -p $O
export HOT_PROF_TOP=${HOT_PROF_TOP:-12}
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py -q -m gpu -x -k "(smoothers or vcycle or iterates or fixed_iterations) and not C3 and not C4 and not C5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== C2 step, default (chained 32-row sub-blocks on level 1)"; timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof_default.log"
echo "== soak, no profiling"; timeout 300 python tools/soak.py C2 8 2>&1 | grep -v amdgpu | tail -3 | tee $O/soak.log
for W in 2048 8192 16384; do echo "== off-block waves $W"; HOT_AMD_AB=1 HOT_GS_OFF_WAVES=$W HOT_PROF_TOP=4 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof_w$W.log"; done
