#!/bin/bash
# Round 4, last collection (one gpurun call): MSDeformAttn parity tests, the default bench line, rocprofv3 kernel trace of the default command.
O=gpurun_out/collect4c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 240 python3 -m pytest tests/test_msda_gpu.py -x -q -m gpu > $O/test_msda.log 2>&1 || { tail -20 $O/test_msda.log; exit 1; }
tail -1 $O/test_msda.log
timeout -k 10 500 python3 bench.py > $O/bench_line_graph_2clips.json 2> $O/bench_graph.err || { tail -5 $O/bench_graph.err; exit 1; }
echo "[1] default bench done"; cut -c1-300 $O/bench_line_graph_2clips.json
rm -rf /tmp/prof_main
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/bench_under_rocprof.log 2>&1 || exit 1
F=$(find /tmp/prof_main -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py $F $O/bench_steady_state_per_step.csv 3 k_scatter_col 4 > $O/steady_summary.txt
cp $(find /tmp/prof_main -name "*kernel_stats.csv" | head -1) $O/bench_rocprofv3_kernel_stats.csv
echo "[2] rocprofv3 of the default command done"; cat $O/steady_summary.txt; grep "k_scatter_col4" $O/bench_steady_state_per_step.csv | cut -c1-40,140-300
