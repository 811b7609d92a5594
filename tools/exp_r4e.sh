#!/bin/bash
O=gpurun_out/exp_r4e; mkdir -p $O
export HOT_PROF_TOP=${HOT_PROF_TOP:-10}
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "generations_agree and C2" > $O/pytest_gen.log 2>&1; echo "generations rc=$?"; tail -5 $O/pytest_gen.log
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py -q -m gpu -x -k "(smoothers or vcycle or iterates or fixed_iterations) and not C3 and not C4 and not C5" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "gs_sub_block=32" > $O/pytest_var.log 2>&1; echo "variants rc=$?"; tail -3 $O/pytest_var.log
echo "== C2 step, fused colour launches"; timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof_fused.log"
echo "== C2 step, two launches per colour"; HOT_AMD_AB=1 HOT_GS_PAIR_UNFUSED=1 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof_unfused.log"
echo "== soak fused"; timeout 300 python tools/soak.py C2 8 2>&1 | grep -v amdgpu | tail -2 | tee $O/soak.log
echo "== soak unfused"; HOT_AMD_AB=1 HOT_GS_PAIR_UNFUSED=1 timeout 300 python tools/soak.py C2 8 2>&1 | grep -v amdgpu | tail -2 | tee $O/soak_unfused.log
