#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box:  bash profiles/run_profiles.sh <tag>   (e.g. r01)
# Writes into gpurun_out/<tag>/; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu"

# 1. the bench line itself (with CPU baseline)
timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"   # the driver's shape
tail -c 3000 "$OUT/bench.json"

# 2. kernel trace + stats of the same command
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err")
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/kt" -name "*kernel_trace.csv" -delete

# 3. HBM traffic counters, one pass each (never combined with other trace domains)
for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err")
done
python profiles/summarize_pmc.py "$OUT/pmc_summary.json" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
find "$OUT" -name "*counter_collection.csv" -delete
du -sh "$OUT"
