#!/bin/bash
# A/B on one box: persistent PCG waited for by the host after every solve (HOT_CG_WAIT, rounds 4 - 6) against not waited for
for rep in 1 2; do
for s in "HOT_X=0" "HOT_CG_WAIT=1"; do
  for c in C2 C1; do
    env HOT_AMD_AB=1 $s timeout 300 python bench.py --config $c --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$s $c', round(d['value'],4), round(d['ms_per_step'],2), d['iterations_per_step'])"
  done
done
done
