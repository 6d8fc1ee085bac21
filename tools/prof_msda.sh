#!/bin/bash
# usage: tools/prof_msda.sh <tag> [ENV=VAL ...]  -- rocprofv3 kernel stats of tools/bench_msda.py into gpurun_out/
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 tools/bench_msda.py > gpurun_out/msda_prof_$tag.log 2>&1 || exit 1
S=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/msda_kernel_stats_$tag.csv
head -12 gpurun_out/msda_kernel_stats_$tag.csv | cut -c1-200
