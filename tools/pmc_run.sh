#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" <kernel-regex> [ENV=VAL ...] -- one rocprofv3 --pmc pass (kernel-trace only) of
# tools/bench_msda.py; per-kernel averages of every counter into gpurun_out/pmc_<tag>.txt
tag=$1; ctrs=$2; kre=$3; shift 3
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 tools/bench_msda.py > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
F=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$F" "$kre" > gpurun_out/pmc_$tag.txt <<'PY'
import csv, re, sys, collections
f, kre = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if kre.search(k):
        mm = re.search(r"(\w+<[^>]*>|\w+)\(", k.replace("(anonymous namespace)::", ""))
        short = mm.group(1) if mm else k[:60]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} avg {sum(v)/len(v):16.1f}   n={len(v)}")
PY
cat gpurun_out/pmc_$tag.txt
