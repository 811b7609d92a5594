#!/bin/bash
# Builds hot_amd/csrc/libhotmi355x_gsclk.so: the product objects with mg_solve.hip compiled -DHOT_GS_CLOCKS (per-pass, per-phase shader clocks of
# the chained GS sweep k_gs_sweep on stderr after every forward sweep).  On the GPU box: HOT_LIB=hot_amd/csrc/libhotmi355x_gsclk.so python tools/vcycle_time.py C2
set -e
cd "$(dirname "$0")/../hot_amd/csrc"
make -s libhotmi355x.so
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result -Wno-unused-value -Wno-pass-failed -DHOT_GS_CLOCKS $GS_EXTRA"
/opt/rocm/bin/hipcc $F -c mg_solve.hip -o /tmp/mg_solve_clk.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "mg_solve.o") /tmp/mg_solve_clk.o -o libhotmi355x_gsclk.so
echo built libhotmi355x_gsclk.so
