#!/bin/bash
mkdir -p gpurun_out/r6f
O=gpurun_out/r6f
timeout 1200 python -m pytest tests/test_gpu_solver.py -x -q -m gpu > $O/t_solver.log 2>&1; echo "solver rc=$?"; grep -v amdgpu.ids $O/t_solver.log | tail -4
timeout 1500 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "HOT_CG or FAKE_TIMEOUT" > $O/t_var.log 2>&1; echo "variants rc=$?"; grep -v amdgpu.ids $O/t_var.log | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "fixed_iterations" > $O/t_full.log 2>&1; echo "full rc=$?"; grep -v amdgpu.ids $O/t_full.log | tail -3
export HOT_PROF_TOP=40
AB=hot_amd/csrc/libhotmi355x_ab.so
for S in "HOT_CG_WGS=256" "HOT_CG_STREAM=1"; do
  echo "== $S"
  env HOT_LIB=$AB $S timeout 300 python tools/prof_table.py C2 2>&1 | grep -E "wall|cg_pers"
done
