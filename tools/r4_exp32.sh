#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 200 python3 tools/bench_text_gate.py 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/prof_tg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tg -- python3 tools/bench_text_gate.py > /dev/null 2>&1
S=$(find /tmp/prof_tg -name "*kernel_stats.csv" | head -1); python3 - $S <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("%-90s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
