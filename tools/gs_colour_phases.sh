#!/bin/bash
# Builds hot_amd/csrc/libhotmi355x_gscclk.so (the product objects with mg_solve.hip compiled -DHOT_GSC_CLOCKS: per-role clocks of k_gs_colour, printed per launch)
# and runs V-cycles on one built C2 hierarchy with it.  usage (GPU box): bash tools/gs_colour_phases.sh [config]
set -e
tools/variant.sh libhotmi355x_gscclk.so mg_solve.hip "-DHOT_GSC_CLOCKS" >/dev/null
mkdir -p gpurun_out/gscclk
env HOT_LIB=hot_amd/csrc/libhotmi355x_gscclk.so HOT_PROF_TOP=4 timeout 300 python tools/vcycle_time.py ${1:-C2} > gpurun_out/gscclk/vcycle.txt 2> gpurun_out/gscclk/clocks.txt || true
tail -n 45 gpurun_out/gscclk/clocks.txt
