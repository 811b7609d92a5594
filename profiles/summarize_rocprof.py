#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats CSV:
   python profiles/summarize_rocprof.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
                            max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.workgroup_size_x)
                     from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                     group by s.kernel_name order by 3 desc""").fetchall()
total = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPRs", "AGPRs", "SGPRs", "LDS_bytes", "WorkgroupSize"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.3f" % (100.0 * r[2] / total)] + list(r[6:]))
print("wrote", sys.argv[2], "kernels:", len(rows), "total kernel ms:", total / 1e6)
