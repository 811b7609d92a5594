#!/bin/bash
# usage: gpu_retry.sh <timeout_s> '<command>'  — retries while gpurun reports "no slot / no box" (exit 3), nothing charged in that case
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_last.txt 2>&1; rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.txt || [ $rc -eq 3 ]; then sleep 45; continue; fi
  break
done
cat /tmp/gpurun_last.txt
