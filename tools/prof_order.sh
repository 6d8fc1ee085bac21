cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_o
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/prof_o.log 2>&1 || exit 1
F=$(find /tmp/prof_o -name "*kernel_trace.csv" | head -1)
python tools/dump_step_order.py $F gpurun_out/step_order.txt
python tools/summarize_trace.py $F gpurun_out/steady_r03b.csv 3 k_scatter_col 4
