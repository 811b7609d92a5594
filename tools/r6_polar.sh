#!/bin/bash
# round 6: k_state with Newton's polar iteration instead of the SVD — parity subset, then the per-record table of C2 / C1 steps
mkdir -p gpurun_out/polar
O=gpurun_out/polar
timeout 1200 python -m pytest tests/test_gpu_force.py tests/test_gpu_golden.py tests/test_gpu_solver.py -x -q -m gpu > $O/t1.log 2>&1; echo "force/golden/solver rc=$?"; tail -3 $O/t1.log
for l in "" _wpe4; do
echo "== lib$l"
HOT_LIB=hot_amd/csrc/libhotmi355x$l.so HOT_PROF_TOP=12 timeout 300 python tools/prof_table.py C2 > $O/prof_C2$l.txt 2>&1; grep "wall\|state_update\|force_scatter" $O/prof_C2$l.txt
HOT_LIB=hot_amd/csrc/libhotmi355x$l.so HOT_PROF_TOP=6 timeout 300 python tools/prof_table.py C1 > $O/prof_C1$l.txt 2>&1; grep "wall\|state_update" $O/prof_C1$l.txt
HOT_LIB=hot_amd/csrc/libhotmi355x$l.so HOT_PROF_TOP=12 timeout 600 python tools/prof_table.py C4 > $O/prof_C4$l.txt 2>&1; grep "wall\|state_update" $O/prof_C4$l.txt
done
