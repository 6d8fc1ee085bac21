#!/bin/bash
# Steps [1]-[3] of tools/collect_profiles.sh (bench lines of the default command and its variants + rocprofv3 steady-state cut) as their own call.
O=gpurun_out/collect; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > $O/bench_line_graph_2clips.json 2> $O/bench_graph.err || exit 1
echo "[1] default bench done"
timeout -k 10 300 python bench.py --eager --no-cpu-baseline > $O/bench_line_eager_2clips.json 2> $O/bench_eager.err || exit 1
timeout -k 10 300 python bench.py --clips-per-gpu 1 --no-cpu-baseline > $O/bench_line_graph_1clip.json 2> $O/bench_1clip.err || exit 1
OCPG_GRAPH_SEGMENTS=3 timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-b1 > $O/bench_line_graph_3segments.json 2> $O/bench_3seg.err || exit 1
echo "[2] eager / 1-clip lines done"
rm -rf /tmp/prof_main
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/bench_under_rocprof.log 2>&1 || exit 1
F=$(find /tmp/prof_main -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $F $O/bench_steady_state_per_step.csv 3 k_scatter_col 4 > $O/steady_summary.txt
cp $(find /tmp/prof_main -name "*kernel_stats.csv" | head -1) $O/bench_rocprofv3_kernel_stats.csv
echo "[3] rocprofv3 of the default command done"; cat $O/steady_summary.txt
