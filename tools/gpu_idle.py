"""Timing aid: how much of a steady-state time step the GPU is idle between kernels, and after which kernels the gaps sit.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu
  python tools/gpu_idle.py /tmp/kt [k]
A Hessian launch marks the start of a step's solve; the k-th one (default: the last timed step = the one before bench.py's profiled
steps) up to the following G2P is analysed; a one-line summary is printed for every step."""
import csv, glob, sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hs = [i for i, r in enumerate(rows) if "k_hessian_rows" in r["Kernel_Name"]]
def bounds(lo):
    g2p = [i for i, r in enumerate(rows) if i > lo and "k_g2p" in r["Kernel_Name"]]
    return lo, (g2p[0] if g2p else len(rows) - 1)
for n, h in enumerate(hs):
    a, b = bounds(h)
    bs = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a : b + 1])
    sp = int(rows[b]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    print("step %d: %d kernels, span %.2f ms, kernel time %.2f ms" % (n, b - a + 1, sp / 1e6, bs / 1e6))
lo, hi = bounds(hs[int(sys.argv[2]) if len(sys.argv) > 2 else 3])
seg = rows[lo : hi + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = 0
gaps = defaultdict(lambda: [0, 0.0])
big = []
prev_end = None
prev_name = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    if prev_end is not None:
        g = max(0, s - prev_end)
        gaps[prev_name + " -> " + name][0] += 1
        gaps[prev_name + " -> " + name][1] += g / 1e3
        if g > 30e3:
            big.append((g / 1e3, prev_name, name))
        busy += e - max(s, prev_end)
    else:
        busy += e - s
    prev_end = max(prev_end or 0, e)
    prev_name = name
span = (t1 - t0) / 1e6
print("span %.2f ms, kernels %d, busy %.2f ms (%.1f %%), idle %.2f ms" % (span, len(seg), busy / 1e6, 100 * busy / 1e6 / span, span - busy / 1e6))
print("gaps by transition (top 25 by total):")
for k, (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %8.1f us total  %5d x  %6.1f us avg   %s" % (tot, n, tot / n, k))
print("gaps > 30 us: %d, sum %.1f us" % (len(big), sum(b[0] for b in big)))
for g, a, b in sorted(big, reverse=True)[:15]:
    print("  %8.1f us  %s -> %s" % (g, a, b))
