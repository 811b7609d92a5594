#!/bin/bash
O=gpurun_out/exp_r4l; mkdir -p $O
export HOT_PROF_TOP=${HOT_PROF_TOP:-10}
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -q -m gpu -x -k "(smoothers or vcycle or iterates or fixed_iterations or generations or rank) and not C3 and not C4 and not C5 and not c4_size" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== C2 step"; timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof.log"
