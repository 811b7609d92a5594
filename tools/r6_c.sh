#!/bin/bash
AB=hot_amd/csrc/libhotmi355x_ab.so
timeout 600 python tools/r6_dbg.py 2>&1 | grep -v amdgpu.ids | grep "p2g"
echo "== warm, presteps 3: stream (clock build)"
HOT_P2G_ONLY=1 HOT_PRESTEPS=3 HOT_LIB=hot_amd/csrc/libhotmi355x_clk.so timeout 300 python tools/p2g_time.py C2 2>&1 | grep -v amdgpu.ids | grep -i p2g | tail -2
for c in "" "HOT_COLD=1"; do for k in "" "HOT_P2G_CELLS2=1"; do
echo "== presteps 3: $c $k"
env $c $k HOT_P2G_ONLY=1 HOT_PRESTEPS=3 HOT_LIB=$AB timeout 300 python tools/p2g_time.py C2 C3 2>&1 | grep -v amdgpu.ids | tail -2
done; done
