#!/bin/bash
O=gpurun_out/exp_r4i; mkdir -p $O
export HOT_PROF_TOP=${HOT_PROF_TOP:-8}
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py -q -m gpu -x -k "(smoothers or vcycle or iterates or fixed_iterations or generations) and not C3 and not C4 and not C5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "gs_sub_block=32" > $O/pytest_var.log 2>&1; echo "variants rc=$?"; tail -3 $O/pytest_var.log
echo "== C2 step"; timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof.log"
echo "== vcycles"; timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | tee $O/vc.log
