#!/bin/bash
O=gpurun_out/exp_r4h; mkdir -p $O
for L in "" hot_amd/csrc/libhotmi355x_x.so; do echo "== lib ${L:-default}"; HOT_LIB=$L HOT_COLD=1 HOT_PRESTEPS=2 timeout 300 python tools/p2g_time.py C2 C3 2>&1 | grep -v amdgpu | tee -a $O/p2g.log; done
