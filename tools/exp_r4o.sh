#!/bin/bash
O=gpurun_out/exp_r4o; mkdir -p $O
for E in "X=1" "HOT_AMD_AB=1 HOT_GS_V1=1" "X=1"; do
  echo "== $E"; env $E timeout 500 python bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','iterations_per_step')}); print([{k:r[k] for k in ('ms_sort','ms_p2g','ms_hessian','ms_mg_build','ms_solve','ms_total')} for r in d['last_step_by_rank']])"
done
