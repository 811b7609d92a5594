#!/bin/bash
mkdir -p gpurun_out/r6e
O=gpurun_out/r6e
bash profiles/run_calibration.sh r06 2>&1 | tail -30
export HOT_PROF_TOP=30
HOT_GS_PROF_COLOURS=1 HOT_LIB=hot_amd/csrc/libhotmi355x_ab.so timeout 300 python tools/prof_table.py C2 > $O/prof_colours.txt 2>&1; grep -E "wall|fused" $O/prof_colours.txt
