#!/bin/bash
# round 6: k_gs_colour with 2 / 3 / 4 wavefronts per workgroup (one substitutes, the others sum the block's previous-colour slots) — per-pass times
mkdir -p gpurun_out/cw
O=gpurun_out/cw
export HOT_PROF_TOP=60 HOT_GS_PROF_COLOURS=1
for l in ${VARIANTS:-ab gscw2 gscw3}; do
  echo "== $l"
  env HOT_LIB=hot_amd/csrc/libhotmi355x_$l.so timeout 300 python tools/vcycle_time.py C2 > $O/${l}.txt 2>&1
  grep "fused" $O/${l}.txt | sort | awk '{printf "%s %s | ", $1, $NF} END {print ""}'; grep fused $O/${l}.txt | awk '{s+=$(NF-3)} END {print "  sum ms/vcycle", s}'
done
