#!/bin/bash
# usage: tools/variant_ab.sh OUT.so FILE.hip "-DFLAG ..."  — the A/B library (HOT_AB_KERNELS) with ONE source recompiled with extra flags
set -e
cd "$(dirname "$0")/../hot_amd/csrc"
make -s libhotmi355x_ab.so
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result -Wno-unused-value -w -DHOT_AB_KERNELS $3"
b=$(basename "$2" .hip)
/opt/rocm/bin/hipcc $F -c "$2" -o "/tmp/${b}_$$.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls ab/*.o | grep -v "^ab/$b.o$") "/tmp/${b}_$$.o" -o "$1"
rm -f "/tmp/${b}_$$.o"
echo built "$1"
