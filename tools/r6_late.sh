#!/bin/bash
# round 6: k_gs_colour with the previous colour's sums taken late — parity subset, then per-pass times
mkdir -p gpurun_out/late
O=gpurun_out/late
timeout 900 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "smoothers or vcycle or iterates" > $O/t_solver.log 2>&1; echo "solver rc=$?"; tail -3 $O/t_solver.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "generations" > $O/t_gen.log 2>&1; echo "generations rc=$?"; tail -3 $O/t_gen.log
export HOT_PROF_TOP=60 HOT_GS_PROF_COLOURS=1
for v in ab; do
  env HOT_LIB=hot_amd/csrc/libhotmi355x_$v.so timeout 300 python tools/vcycle_time.py C2 > $O/$v.txt 2>&1
  grep "fused" $O/$v.txt | sort | awk '{printf "%s %s | ", $1, $NF} END {print ""}'; grep fused $O/$v.txt | awk '{s+=$(NF-3)} END {print "  sum ms/vcycle", s}'
done
unset HOT_GS_PROF_COLOURS
HOT_PROF_TOP=12 timeout 300 python tools/prof_table.py C2 > $O/prof_prod.txt 2>&1; head -14 $O/prof_prod.txt
