#!/bin/bash
O=gpurun_out/exp_r4k; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_force.py tests/test_gpu_golden.py tests/test_gpu_transfer.py tests/test_gpu_solver.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo "== transfers / state"; HOT_COLD=1 HOT_PRESTEPS=2 timeout 300 python tools/p2g_time.py C2 C3 C4 2>&1 | grep -v amdgpu | tee $O/p2g.log
echo "== C2 step"; HOT_PROF_TOP=12 timeout 300 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof.log"
