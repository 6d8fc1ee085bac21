#!/bin/bash
# usage: tools/pmc_any.sh <tag> <kernel-regex> <script.py> COUNTER [COUNTER ...]  -- one rocprofv3 --pmc pass per counter (kernel-trace
# only) of `python3 <script.py>`; per-kernel averages into gpurun_out/pmc_<tag>.txt
tag=$1; kre=$2; script=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/pmc_$tag.txt
for c in "$@"; do
  rm -rf /tmp/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$c -- python3 $script > gpurun_out/pmc_${tag}_$c.log 2>&1 || { echo "$c: pass failed" >> gpurun_out/pmc_$tag.txt; continue; }
  F=$(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$kre" >> gpurun_out/pmc_$tag.txt <<'PY'
import csv, re, sys, collections
f, kre = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if kre.search(row["Kernel_Name"]):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"{c:32s} avg {sum(v)/len(v):18.1f}   n={len(v)}")
PY
done
cat gpurun_out/pmc_$tag.txt
