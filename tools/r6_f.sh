#!/bin/bash
mkdir -p gpurun_out/r6f
O=gpurun_out/r6f
timeout 1200 python -m pytest tests/test_gpu_solver.py -x -q -m gpu > $O/t_solver.log 2>&1; echo "solver rc=$?"; grep -v amdgpu.ids $O/t_solver.log | tail -4
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "generations or fixed_iterations or (invariants and C2) or (invariants and C3)" > $O/t_full.log 2>&1; echo "full rc=$?"; grep -v amdgpu.ids $O/t_full.log | tail -4
timeout 1500 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "HESSIAN" > $O/t_var.log 2>&1; echo "variants rc=$?"; grep -v amdgpu.ids $O/t_var.log | tail -3
export HOT_PROF_TOP=40
timeout 300 python tools/prof_table.py C2 > $O/prof_prod.txt 2>&1; grep -E "wall|hessian|gs_images|mg_RAP|mg_AP|gs_split|gs_winv|fused" $O/prof_prod.txt
timeout 300 python tools/prof_table.py C3 > $O/prof_C3.txt 2>&1; grep -E "wall|hessian|gs_images|p2g|fused" $O/prof_C3.txt
