#!/bin/bash
O=gpurun_out/exp_r4m; mkdir -p $O
export HOT_PROF_TOP=20 HOT_AMD_AB=1
for D in 8 4 6 10 12 16; do echo "== substitution depth $D"; HOT_GS_SUBST_D=$D timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | grep "gs_forward_L0\|gs_backward_L0" | tee -a $O/vc.log; done
for D in 8 16; do echo "== depth $D, kernel pair on level 1 too"; HOT_SOAK_CFG=gs_sub_block=32,gs_chain=1 HOT_GS_SUBST_D=$D timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | grep "gs_forward_L1\|gs_backward_L1\|_off_L1" | tee -a $O/vc.log; done
