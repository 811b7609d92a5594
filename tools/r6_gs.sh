#!/bin/bash
# round 6: the one-launch colour pass (k_gs_colour) — parity subset, then per-record tables of C2 steps for the production build and the A/B switches
mkdir -p gpurun_out/r6gs
O=gpurun_out/r6gs
timeout 900 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "smoothers or vcycle or iterates" > $O/t_solver.log 2>&1; echo "solver rc=$?"; tail -3 $O/t_solver.log
timeout 1500 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "gs_chain=1 and gs_sub_block=32" > $O/t_variants.log 2>&1; echo "variants rc=$?"; tail -3 $O/t_variants.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "generations" > $O/t_gen.log 2>&1; echo "generations rc=$?"; tail -3 $O/t_gen.log
export HOT_PROF_TOP=14
timeout 300 python tools/prof_table.py C2 > $O/prof_prod.txt 2>&1; head -16 $O/prof_prod.txt
AB=hot_amd/csrc/libhotmi355x_ab.so
for S in "HOT_GS_PAIR=1" "HOT_GS_SUBST_D=4" "HOT_GS_SUBST_D=6" "HOT_GS_OFF_WAVES=8192" "HOT_GS_OFF_WAVES=2048"; do
  echo "== $S"
  env HOT_LIB=$AB $S timeout 300 python tools/prof_table.py C2 > "$O/prof_$S.txt" 2>&1; head -9 "$O/prof_$S.txt"
done
