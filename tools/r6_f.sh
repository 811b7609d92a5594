#!/bin/bash
mkdir -p gpurun_out/r6f
O=gpurun_out/r6f
timeout 1200 python -m pytest tests/test_gpu_solver.py -x -q -m gpu > $O/t_solver.log 2>&1; echo "solver rc=$?"; grep -v amdgpu.ids $O/t_solver.log | tail -5
timeout 1500 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "PAIR or SUBST_D or TURN or OFF_WAVES" > $O/t_var.log 2>&1; echo "variants rc=$?"; grep -v amdgpu.ids $O/t_var.log | tail -4
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "generations or fixed_iterations" > $O/t_full.log 2>&1; echo "full rc=$?"; grep -v amdgpu.ids $O/t_full.log | tail -5
export HOT_PROF_TOP=40
timeout 300 python tools/prof_table.py C2 > $O/prof_prod.txt 2>&1; grep -v amdgpu.ids $O/prof_prod.txt | head -8
AB=hot_amd/csrc/libhotmi355x_ab.so
for S in "HOT_GS_SUBST_D=8" "HOT_GS_NO_TURN=1"; do
  echo "== $S"
  env HOT_LIB=$AB $S timeout 300 python tools/prof_table.py C2 > "$O/prof_$S.txt" 2>&1; grep -E "wall|fused" "$O/prof_$S.txt" | head -18
done
HOT_GS_PROF_COLOURS=1 HOT_LIB=hot_amd/csrc/libhotmi355x_ab.so timeout 300 python tools/prof_table.py C2 2>&1 | grep -E "wall|fused"
