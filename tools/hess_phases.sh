#!/bin/bash
# Builds hot_amd/csrc/libhotmi355x_clk.so (the product objects + hessian_tiles.hip with -DHOT_HT_CLOCKS); run `HOT_LIB=... python tools/hess_time.py C2` on the GPU box.
set -e
cd "$(dirname "$0")/../hot_amd/csrc"
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result -Wno-unused-value -DHOT_HT_CLOCKS -DHOT_AB_KERNELS -c hessian_tiles.hip -o /tmp/hessian_tiles_clk.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls ab/*.o | grep -v hessian_tiles.o) /tmp/hessian_tiles_clk.o -o libhotmi355x_clk.so
echo built libhotmi355x_clk.so
