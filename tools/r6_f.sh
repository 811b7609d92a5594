#!/bin/bash
mkdir -p gpurun_out/r6f
O=gpurun_out/r6f
export HOT_PROF_TOP=12
for i in 1 2; do
echo "== NT (product lib)"; timeout 300 python tools/prof_table.py C2 2>&1 | grep -E "wall|fused|residual_L0|apmv_L0"
echo "== before (wt_prev lib)"; HOT_LIB=gpurun_prev/libhotmi355x.so timeout 300 python tools/prof_table.py C2 2>&1 | grep -E "wall|fused|residual_L0|apmv_L0"
done
python - <<'PY'
import hot_amd, bench, numpy as np
from hot_amd import synth, parallel
cfg = dict(synth.CONFIGS["C2"]); cloud = parallel.shard_cloud(cfg, 0, 1, n=10)
ctx = bench.make_ctx(hot_amd.load(), cloud, cfg)
print("copy kernel GB/s", ctx.copy_bandwidth(1 << 30, 20))
PY
