#!/bin/bash
# Round 4, second half (one gpurun call): the MSDeformAttn parity tests, grad_value through both kernel families on ring / noisy / trained-like
# offsets (warm and on cold operands), the default bench line (incl. b1, roofline rows, cpu_baseline), rocprofv3 kernel trace of the default
# command and -- unless QUICK=1 -- the PMC passes of the grad_value entry point.
# Everything lands in gpurun_out/collect4b/; copy what is to be judged into profiles/ as r04_*.
O=gpurun_out/collect4b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 240 python3 -m pytest tests/test_msda_gpu.py -x -q -m gpu > $O/test_msda.log 2>&1 || { tail -20 $O/test_msda.log; exit 1; }
tail -1 $O/test_msda.log
for cold in 0 1; do
  GV_COLD=$cold GV_MODES=ring,ring+n,trained GV_PATHS=1,0 GV_SELECT=1 ITERS=40 timeout -k 10 200 python3 tools/bench_msda_gv.py 2>&1 | grep '^{' | sed "s/^/cold=$cold /" >> $O/msda_gv_paths.txt
done
cat $O/msda_gv_paths.txt
echo "[0] grad_value paths done"
timeout -k 10 500 python3 bench.py > $O/bench_line_graph_2clips.json 2> $O/bench_graph.err || { tail -5 $O/bench_graph.err; exit 1; }
echo "[1] default bench done"; cut -c1-400 $O/bench_line_graph_2clips.json
rm -rf /tmp/prof_main
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/bench_under_rocprof.log 2>&1 || exit 1
F=$(find /tmp/prof_main -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py $F $O/bench_steady_state_per_step.csv 3 k_scatter_col 4 > $O/steady_summary.txt
cp $(find /tmp/prof_main -name "*kernel_stats.csv" | head -1) $O/bench_rocprofv3_kernel_stats.csv
echo "[2] rocprofv3 of the default command done"; cat $O/steady_summary.txt
[ "$QUICK" = "1" ] && exit 0
bash tools/pmc_gv.sh > $O/pmc_gv.log 2>&1 || { tail -5 $O/pmc_gv.log; exit 1; }
cp gpurun_out/r04_msda_pmc.json $O/ 2>/dev/null
for m in ring trained; do for c in f w a; do cp gpurun_out/pmcgv_${m}_$c.csv $O/msda_pmc_rows_${m}_$c.csv; done; done
echo "[3] PMC done"; tail -3 $O/pmc_gv.log
