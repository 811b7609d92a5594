#!/bin/bash
# round 6: k_gs_colour with some of the streaming workgroups dispatched before the block workgroups (HOT_GS_STREAM_FIRST, A/B build) — per-pass times
mkdir -p gpurun_out/front
O=gpurun_out/front
export HOT_PROF_TOP=60 HOT_GS_PROF_COLOURS=1
for f in ${FRONTS:-0 128 256 512 1024}; do
  echo "== stream first $f"
  env HOT_LIB=hot_amd/csrc/libhotmi355x_ab.so HOT_GS_STREAM_FIRST=$f timeout 300 python tools/vcycle_time.py C2 > $O/f$f.txt 2>&1
  grep "fused" $O/f$f.txt | sort | awk '{printf "%s %s | ", $1, $NF} END {print ""}'; grep fused $O/f$f.txt | awk '{s+=$(NF-3)} END {print "  sum ms/vcycle", s}'
done
