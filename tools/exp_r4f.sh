#!/bin/bash
O=gpurun_out/exp_r4f; mkdir -p $O
export HOT_PROF_TOP=3 HOT_AMD_AB=1
for T in 1 8 32 257 264 $((65536*3+8)) $((65536*3+264)); do echo "== tune $T"; HOT_GS_PAIR_TUNE=$T timeout 200 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof_$T.log"; done
for W in 1024 4096; do echo "== tune 264 waves $W"; HOT_GS_PAIR_TUNE=264 HOT_GS_OFF_WAVES=$W timeout 200 python tools/prof_table.py C2 2>&1 | grep -v amdgpu | tee "$O/prof_w$W.log"; done
