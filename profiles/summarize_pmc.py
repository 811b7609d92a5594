#!/usr/bin/env python3
"""Reduce rocprofv3 `--pmc X --output-format csv` counter_collection CSVs to per-kernel averages.

   python profiles/summarize_pmc.py <out.json> <dir-or-csv> [<dir-or-csv> ...]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B; on gfx950 FETCH_SIZE counts 128-B
requests as 64 B (MI355X_MICROARCH.md, "HBM"), so the read bytes are doubled here.  WRITE_SIZE is taken as is
(uncalibrated, the guide says so).  Output: {kernel: {counter: {"avg": .., "sum": .., "n": ..}, "hbm_bytes_per_launch": ..}}.
"""
import csv
import glob
import json
import os
import sys

out, srcs = sys.argv[1], sys.argv[2:]
files = []
for s in srcs:
    files += [s] if os.path.isfile(s) else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
acc = {}
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name") or row.get("kernel_name")
            c = row.get("Counter_Name") or row.get("counter_name")
            v = float(row.get("Counter_Value") or row.get("counter_value") or 0)
            a = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
            a[0] += v
            a[1] += 1
res = {}
for k, cs in acc.items():
    r = {c: {"avg": a[0] / a[1], "sum": a[0], "n": a[1]} for c, a in cs.items()}
    if "FETCH_SIZE" in r or "WRITE_SIZE" in r:
        rd = 2.0 * 1024.0 * r.get("FETCH_SIZE", {"avg": 0})["avg"]
        wr = 1024.0 * r.get("WRITE_SIZE", {"avg": 0})["avg"]
        r["hbm_read_bytes_per_launch"] = rd
        r["hbm_write_bytes_per_launch"] = wr
        r["hbm_bytes_per_launch"] = rd + wr
    res[k] = r
with open(out, "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
print("wrote", out, "kernels:", len(res), "from", len(files), "files")
