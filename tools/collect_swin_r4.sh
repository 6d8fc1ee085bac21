#!/bin/bash
# the Swin half of tools/collect_profiles_r4.sh (configs #4 / #5: bench lines + rocprofv3 traces)
O=gpurun_out/collect4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --backbone video_swin_t_p4w7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_line_swint.json 2> $O/bench_swint.err || { tail -5 $O/bench_swint.err; exit 1; }
rm -rf /tmp/prof_swint
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swint -- python3 bench.py --backbone video_swin_t_p4w7 --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swint_under_rocprof.log 2>&1 || exit 1
cp $(find /tmp/prof_swint -name "*kernel_stats.csv" | head -1) $O/swint_rocprofv3_kernel_stats.csv
python3 tools/summarize_trace.py $(find /tmp/prof_swint -name "*kernel_trace.csv" | head -1) $O/swint_steady_state_per_step.csv 2 k_scatter_col 4
timeout -k 10 500 python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_line_swinb_roberta_fp16.json 2> $O/bench_swinb.err || { tail -5 $O/bench_swinb.err; exit 1; }
rm -rf /tmp/prof_swinb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swinb -- python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swinb_under_rocprof.log 2>&1 || exit 1
cp $(find /tmp/prof_swinb -name "*kernel_stats.csv" | head -1) $O/swinb_roberta_fp16_rocprofv3_kernel_stats.csv
python3 tools/summarize_trace.py $(find /tmp/prof_swinb -name "*kernel_trace.csv" | head -1) $O/swinb_roberta_fp16_steady_state_per_step.csv 2 k_scatter_col 4
