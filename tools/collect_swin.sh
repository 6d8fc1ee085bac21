O=gpurun_out/collect; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --backbone video_swin_t_p4w7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_line_swint.json 2> $O/bench_swint.err || { tail -5 $O/bench_swint.err; exit 1; }
rm -rf /tmp/prof_swint
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swint -- python bench.py --backbone video_swin_t_p4w7 --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swint_under_rocprof.log 2>&1 || exit 1
cp $(find /tmp/prof_swint -name "*kernel_stats.csv" | head -1) $O/swint_rocprofv3_kernel_stats.csv
python tools/summarize_trace.py $(find /tmp/prof_swint -name "*kernel_trace.csv" | head -1) $O/swint_steady_state_per_step.csv 2 k_scatter_col 4 > $O/swint_steady_summary.txt; cat $O/swint_steady_summary.txt
echo "[5] Swin-T done"
bash tools/collect_swinb.sh
