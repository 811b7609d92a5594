#!/bin/bash
export HOT_PROF_TOP=4 HOT_AMD_AB=1
for W in 4096 2048 3072 4096 2048 3072; do echo "== off-block waves $W"; HOT_GS_OFF_WAVES=$W timeout 200 python tools/vcycle_time.py C2 2>&1 | grep -v amdgpu | grep "_off_L0"; done
