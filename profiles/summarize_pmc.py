#!/usr/bin/env python3
"""Reduce rocprofv3 `--pmc X --output-format csv` counter_collection CSVs to per-kernel averages.

   python profiles/summarize_pmc.py <out.json> <dir-or-csv> [<dir-or-csv> ...]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B; on gfx950 FETCH_SIZE counts 128-B
requests as 64 B (MI355X_MICROARCH.md, "HBM"), so the read bytes are doubled here.  WRITE_SIZE is taken as is
(uncalibrated, the guide says so).  Output: {kernel: {counter: {"avg": .., "sum": .., "n": ..}, "hbm_bytes_per_launch": ..}},
plus "_build": {"source_sha16": ..} — a hash of hot_amd/csrc/*.hip and *.h as they are when the summary is made, which bench.py
compares with the tree it runs from (file times do not survive a checkout).
"""
import csv
import glob
import hashlib
import json
import os
import sys


def csrc_sha16(root):
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "hot_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "hot_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


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
res["_build"] = {"source_sha16": csrc_sha16(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}
with open(out, "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
print("wrote", out, "kernels:", len(res), "from", len(files), "files")
