#!/bin/bash
# PMC calibration (round 6): FETCH_SIZE / WRITE_SIZE of two kernels whose bytes are known exactly, one counter per pass.
#   bash profiles/run_calibration.sh r06   -> gpurun_out/<tag>/pmc_calibration.json (copy it to profiles/<tag>_pmc_calibration.json)
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/cal_$C" -o cal -- python $ROOT/tools/pmc_calib.py > "$OUT/cal_$C.json" 2> "$OUT/cal_$C.err")
done
python profiles/summarize_pmc.py "$OUT/cal_summary.json" "$OUT/cal_FETCH_SIZE" "$OUT/cal_WRITE_SIZE"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
known = json.loads([l for l in open(out + "/cal_FETCH_SIZE.json") if l.startswith("{")][-1])
summ = json.load(open(out + "/cal_summary.json"))
res = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_calib.py; counters in KiB, FETCH_SIZE x 2 (gfx950 counts a 128-byte request as 64 bytes), as profiles/summarize_pmc.py applies it to every kernel",
       "_build": summ.get("_build")}
for key, pat in (("k_spmv_L0", "k_spmv"), ("k_copy16", "k_copy16")):
    rec = [v for k, v in summ.items() if pat in k]
    if not rec:
        continue
    r = rec[0]
    kn = known[key]
    res[key] = {"known_read_bytes": kn["read_bytes"], "known_write_bytes": kn["write_bytes"], "pmc_read_bytes_x2": r.get("hbm_read_bytes_per_launch"), "pmc_write_bytes": r.get("hbm_write_bytes_per_launch"),
                "read_ratio_pmc_over_known": r.get("hbm_read_bytes_per_launch", 0) / kn["read_bytes"], "write_ratio_pmc_over_known": r.get("hbm_write_bytes_per_launch", 0) / kn["write_bytes"],
                "launches_counted": r.get("FETCH_SIZE", {}).get("n"), **{k: v for k, v in kn.items() if k not in ("read_bytes", "write_bytes")}}
json.dump(res, open(out + "/pmc_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*counter_collection.csv" -delete
