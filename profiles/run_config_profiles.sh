#!/bin/bash
# Bench line + rocprofv3 kernel stats of the other BASELINE configurations at full size on one MI355X:
#   bash profiles/run_config_profiles.sh <tag> C1 C3 C4 C5     -> gpurun_out/<tag>/bench_<cfg>.json, kernel_stats_<cfg>.csv
set -u
TAG=${1:-r03}
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in "$@"; do
    timeout 1500 python bench.py --config $C --no-cpu --steps 2 --warmup 1 > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
    echo "$C bench rc=$?"
    (cd /tmp && timeout 1800 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$C" -o kt -- python $ROOT/bench.py --config $C --no-cpu --steps 2 --warmup 1 > "$OUT/bench_${C}_under_rocprof.json" 2> "$OUT/kt_$C.err")
    find "$OUT/kt_$C" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$C.csv" \;
    rm -rf "$OUT/kt_$C"
    head -8 "$OUT/kernel_stats_$C.csv" | cut -c1-160
done
