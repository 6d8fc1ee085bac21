"""Diagnostic: the kernels of ONE steady-state bench step in launch order (name, duration) from a rocprofv3 kernel trace.
usage: dump_step_order.py <kernel_trace.csv> <out.txt> [marker] [markers_per_step]"""
import csv
import sys
path, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "k_scatter_col"
per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
lo, hi = marks[-per_step - 1] + 1, marks[-1] + 1
prev = None
with open(out, "w") as f:
    for s, e, name in rows[lo:hi]:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        short = name.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
        f.write("%7.1f us  gap %6.1f  %s\n" % ((e - s) / 1e3, gap, short[:150]))
        prev = e
