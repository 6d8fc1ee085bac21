#!/bin/bash
# usage: tools/prof_py.sh <tag> <script.py> [ENV=VAL ...] -- rocprofv3 kernel stats of a python script into gpurun_out/kstats_<tag>.csv
tag=$1; script=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $script > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
S=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kstats_$tag.csv
python3 - <<PY
import csv
for i, r in enumerate(csv.reader(open("gpurun_out/kstats_$tag.csv"))):
    if i == 0 or i > 28: continue
    print(f"{float(r[3])/1e3:9.1f} us x{r[1]:>5s}  {r[0][:110]}")
PY
