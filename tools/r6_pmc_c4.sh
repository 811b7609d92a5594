#!/bin/bash
# round 6: HBM traffic counters of the colour pass at C4 (16 M particles, four levels, the whole body on one GPU) — separate FETCH_SIZE / WRITE_SIZE passes of one step each
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_C4; mkdir -p "$OUT"; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 1500 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python $ROOT/bench.py --config C4 --steps 1 --warmup 0 --no-cpu > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err")
done
python profiles/summarize_pmc.py "$OUT/pmc_summary.json" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
find "$OUT" -name "*counter_collection.csv" -delete
python - <<'P'
import json
d=json.load(open('gpurun_out/pmc_C4/pmc_summary.json'))
for k,v in d.items():
    if 'k_gs_colour' in k or 'k_gs_residual<double, true>' in k: print(k[:60], round(v['hbm_bytes_per_launch']/1e6,1), 'MB', v['FETCH_SIZE']['n'])
b=json.load(open('gpurun_out/pmc_C4/pmc_FETCH_SIZE.json')); r=b['roofline']; print('algorithmic MB per launch', r['algorithmic_bytes_per_launch']/1e6, 'avg us', 1e3*r['avg_launch_ms'])
P
