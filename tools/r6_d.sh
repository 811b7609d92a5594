#!/bin/bash
mkdir -p gpurun_out/r6d
O=gpurun_out/r6d
timeout 1200 python -m pytest tests/test_gpu_solver.py tests/test_abi_load.py -x -q -m gpu -k "pinned or abi or adapter or objective or converge" > $O/t_pin.log 2>&1; echo "pin rc=$?"; grep -v amdgpu.ids $O/t_pin.log | tail -30
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "c1_full" > $O/t_c1.log 2>&1; echo "c1 rc=$?"; grep -v amdgpu.ids $O/t_c1.log | tail -6
timeout 2400 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $O/t_mr.log 2>&1; echo "multirank rc=$?"; grep -v amdgpu.ids $O/t_mr.log | tail -6
bash profiles/run_calibration.sh r06 2>&1 | tail -40
