"""Timing aid: per-dispatch durations of the finest-level GS kernel from a rocprofv3 kernel trace, grouped by position
inside the half sweep (colour x sub-block), to see which passes are the slow ones.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $REPO/tools/dbg_gs.py
  python tools/gs_pass_times.py /tmp/kt"""
import csv, glob, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for tag in ("k_gs_block<double, true, 32>", "k_gs_block<double, false, 32>"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if tag in r["Kernel_Name"]]
    if not d:
        continue
    per = defaultdict(list)
    for i, t in enumerate(d):
        per[i % 16].append(t)
    print(tag, "launches", len(d), "mean %.1f us" % (sum(d) / len(d)))
    print("  by pass:", " ".join("%.1f" % (sum(v) / len(v)) for k, v in sorted(per.items())))
