#!/bin/bash
# Builds hot_amd/csrc/libhotmi355x_clk.so: the product objects with hessian_rows.hip, hessian_tiles.hip and transfer.hip compiled -DHOT_HT_CLOCKS (per-phase
# shader clocks of k_hessian_rows / k_hessian_tiles2 / k_p2g_cells2 on stderr).  On the GPU box: HOT_LIB=hot_amd/csrc/libhotmi355x_clk.so python tools/hess_time.py C2
set -e
cd "$(dirname "$0")/../hot_amd/csrc"
make -s libhotmi355x.so
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result -Wno-unused-value -DHOT_HT_CLOCKS $HT_EXTRA"
/opt/rocm/bin/hipcc $F -c hessian_tiles.hip -o /tmp/hessian_tiles_clk.o &
/opt/rocm/bin/hipcc $F -c hessian_rows.hip -o /tmp/hessian_rows_clk.o &
/opt/rocm/bin/hipcc $F -c transfer.hip -o /tmp/transfer_clk.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "hessian_tiles.o\|hessian_rows.o\|transfer.o") /tmp/hessian_tiles_clk.o /tmp/hessian_rows_clk.o /tmp/transfer_clk.o -o ${HT_OUT:-libhotmi355x_clk.so}
echo built ${HT_OUT:-libhotmi355x_clk.so}
