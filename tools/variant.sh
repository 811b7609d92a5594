#!/bin/bash
# usage: tools/variant.sh OUT.so FILE.hip "-DFLAG ..."  — the product library with ONE source recompiled with extra flags (A/B experiments on
# the GPU box: HOT_LIB=hot_amd/csrc/OUT.so python tools/hess_time.py C2, tools/p2g_time.py ...)
set -e
cd "$(dirname "$0")/../hot_amd/csrc"
make -s libhotmi355x.so
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-result -Wno-unused-value -w $3"
b=$(basename "$2" .hip)
/opt/rocm/bin/hipcc $F -c "$2" -o "/tmp/${b}_$$.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^$b.o$") "/tmp/${b}_$$.o" -o "$1"
rm -f "/tmp/${b}_$$.o"
echo built "$1"
