#!/bin/bash
# round 6: what bounds the block role of k_gs_colour — V-cycles on one C2 hierarchy, one record per colour pass, production A/B build against timing-only variants
mkdir -p gpurun_out/gsexp
O=gpurun_out/gsexp
export HOT_PROF_TOP=60 HOT_GS_PROF_COLOURS=1
for v in ${VARIANTS:-ab gsexp1 gsexp2 gsexp3 gsexp4}; do
  echo "== $v"
  env HOT_LIB=hot_amd/csrc/libhotmi355x_$v.so timeout 300 python tools/vcycle_time.py C2 > $O/$v.txt 2>&1
  grep "fused" $O/$v.txt | sort | awk '{printf "%s %s | ", $1, $NF} END {print ""}'; grep fused $O/$v.txt | awk '{s+=$(NF-3)} END {print "  sum ms/vcycle", s}'
done
