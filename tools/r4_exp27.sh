#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
rm -rf /tmp/prof_dft
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dft -- python3 tools/bench_lfm_dft.py > gpurun_out/r4/lfm_dft_prof.log 2>&1 || { tail -5 gpurun_out/r4/lfm_dft_prof.log; exit 1; }
F=$(find /tmp/prof_dft -name "*kernel_trace.csv" | head -1)
python3 - $F <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "rows_" in n or "cols_" in n:
        key = (n.split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Workgroup_Size_X"))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print(k, "n=%d median %.1f us min %.1f" % (len(v), v[len(v) // 2], v[0]))
PY
