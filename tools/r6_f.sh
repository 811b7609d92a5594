#!/bin/bash
mkdir -p gpurun_out/r6f
O=gpurun_out/r6f
timeout 1200 python -m pytest tests/test_gpu_solver.py -x -q -m gpu > $O/t_solver.log 2>&1; echo "solver rc=$?"; grep -v amdgpu.ids $O/t_solver.log | tail -5
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "fixed_iterations or invariants and not C5 and not C4" > $O/t_full.log 2>&1; echo "full rc=$?"; grep -v amdgpu.ids $O/t_full.log | tail -5
timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "one_body or c2_size" > $O/t_mr.log 2>&1; echo "mr rc=$?"; grep -v amdgpu.ids $O/t_mr.log | tail -4
export HOT_PROF_TOP=16
timeout 300 python tools/prof_table.py C2 > $O/prof_prod.txt 2>&1; grep -v amdgpu.ids $O/prof_prod.txt | head -18
