#!/bin/bash
O=gpurun_out/exp_r4n; mkdir -p $O
export HOT_PROF_TOP=12
timeout 600 python -m pytest tests/test_gpu_solver.py -q -m gpu -x -k "smoothers or vcycle or iterates" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for L in "" hot_amd/csrc/libhotmi355x_x.so "" hot_amd/csrc/libhotmi355x_x.so; do echo "== lib ${L:-default (nt loads)}"; HOT_LIB=$L timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | tee -a $O/vc.log; done
