#!/bin/bash
# usage: tools/prof_step.sh <tag> [ENV=VAL ...]   -- rocprofv3 kernel trace of bench.py, steady-state summary into gpurun_out/
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 $BENCH_ARGS > gpurun_out/bench_prof_$tag.log 2>&1 || exit 1
F=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $F gpurun_out/steady_$tag.csv 3 k_scatter_col 4
S=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/kernel_stats_$tag.csv
