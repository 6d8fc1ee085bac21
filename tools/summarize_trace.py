"""Summarise a rocprofv3 --kernel-trace CSV over the LAST `n` bench steps (steady state, after MIOpen's find phase).

usage: summarize_trace.py <kernel_trace.csv> <out.csv> [n_steps] [marker_substring] [markers_per_step]
A step boundary is recognised by the marker kernel (default: msda_bwd_fast, 8 launches per step = 4 enc + 4 dec).
"""
import csv
import sys
from collections import defaultdict

path, out = sys.argv[1], sys.argv[2]
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
marker = sys.argv[4] if len(sys.argv) > 4 else "msda_bwd_fast"
per_step = int(sys.argv[5]) if len(sys.argv) > 5 else 8

rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
assert len(marks) >= per_step * (n_steps + 1), (len(marks), per_step, n_steps)
# window: from just after the last marker of step (-n-1) to the last marker of the final step
lo = marks[-per_step * n_steps - 1] + 1
hi = marks[-1] + 1
sel = rows[lo:hi]
wall = (sel[-1][1] - sel[0][0]) / 1e6
agg = defaultdict(lambda: [0, 0])
for s, e, name in sel:
    agg[name][0] += e - s
    agg[name][1] += 1
busy = sum(v[0] for v in agg.values()) / 1e6
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# window: last %d steps (marker-to-marker), wall %.3f ms, GPU busy %.3f ms, launches %d" % (n_steps, wall, busy, len(sel))])
    w.writerow(["Name", "Calls", "TotalMs", "AvgUs", "PctOfBusy", "CallsPerStep", "MsPerStep"])
    for name, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([name[:160], c, "%.3f" % (ns / 1e6), "%.2f" % (ns / c / 1e3), "%.2f" % (100 * ns / 1e6 / busy), "%.1f" % (c / n_steps), "%.3f" % (ns / 1e6 / n_steps)])
print("window wall %.2f ms/step, busy %.2f ms/step, %d launches/step" % (wall / n_steps, busy / n_steps, len(sel) // n_steps))
