#!/bin/bash
O=gpurun_out/exp_r4j; mkdir -p $O
export HOT_PROF_TOP=14 HOT_AMD_AB=1
for E in "X=1" "HOT_GS_PASS_COUNTERS=1" "HOT_GS_BLOCK_FLAGS=1"; do echo "== $E"; env $E timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | grep "_L1\|_L2" | tee -a $O/vc.log; done
