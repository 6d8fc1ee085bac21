#!/bin/bash
# usage: tools/prof_any.sh <tag> <script.py> [ENV=VAL ...] -- rocprofv3 kernel stats of a python script; top kernels into gpurun_out/
tag=$1; script=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $script > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
S=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$tag.csv
python3 - gpurun_out/kernel_stats_$tag.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.1f us  min %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
