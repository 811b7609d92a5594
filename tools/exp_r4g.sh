#!/bin/bash
O=gpurun_out/exp_r4g; mkdir -p $O
export HOT_PROF_TOP=4 HOT_AMD_AB=1
echo "== unfused"; HOT_GS_PAIR_UNFUSED=1 timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | tee $O/unfused.log
for T in 288 $((288+512)) $((288+1024)) $((288+1536)); do echo "== tune $T"; HOT_GS_PAIR_TUNE=$T timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | tee "$O/t_$T.log"; done
