#!/bin/bash
echo "== previous commit (worktree)"
(cd wt_prev && timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -s -k "three_time_steps_fp32" 2>&1 | grep -v amdgpu.ids | grep -E "fp32 step|passed|failed" | tail -16)
echo "== HEAD"
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -s -k "three_time_steps_fp32" 2>&1 | grep -v amdgpu.ids | grep -E "fp32 step|passed|failed" | tail -16
echo "== HEAD, again"
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -s -k "three_time_steps_fp32" 2>&1 | grep -v amdgpu.ids | grep -E "fp32 step|passed|failed" | tail -16
