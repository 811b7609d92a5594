#!/bin/bash
# Round-4 experiment batch c: XCD-dealt rows of the SpMV-like kernels (this build) and level-1 launch structures by knob.
O=gpurun_out/exp_r4c; mkdir -p $O
export HOT_PROF_TOP=${HOT_PROF_TOP:-16}
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py -q -m gpu -x -k "(smoothers or vcycle or iterates or fixed_iterations) and not C3 and not C4 and not C5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for K in "" "gs_sub_block=32" "gs_sub_block=32,gs_chain=1" "gs_sub_block=16" "gs_chain=1"; do
  echo "== C2 step, HOT_SOAK_CFG=$K"; HOT_SOAK_CFG=$K timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof_$K.log"
done
