#!/bin/bash
# Round-4 measurement set (one gpurun call): the default bench line (incl. b1, roofline rows, cpu_baseline), rocprofv3 kernel trace of the
# default command, Swin-T (config #4) and Swin-B + RoBERTa fp16 (config #5) lines + traces, the LFM / conv / wgrad micro-benchmarks and
# the glue census.  Everything lands in gpurun_out/collect4/; copy what is to be judged into profiles/ as r04_*.
O=gpurun_out/collect4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py > $O/bench_line_graph_2clips.json 2> $O/bench_graph.err || { tail -5 $O/bench_graph.err; exit 1; }
echo "[1] default bench done"
rm -rf /tmp/prof_main
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_main -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/bench_under_rocprof.log 2>&1 || exit 1
F=$(find /tmp/prof_main -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py $F $O/bench_steady_state_per_step.csv 3 k_scatter_col 4 > $O/steady_summary.txt
cp $(find /tmp/prof_main -name "*kernel_stats.csv" | head -1) $O/bench_rocprofv3_kernel_stats.csv
echo "[2] rocprofv3 of the default command done"; cat $O/steady_summary.txt
timeout -k 10 400 python3 bench.py --backbone video_swin_t_p4w7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_line_swint.json 2> $O/bench_swint.err || { tail -5 $O/bench_swint.err; exit 1; }
rm -rf /tmp/prof_swint
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swint -- python3 bench.py --backbone video_swin_t_p4w7 --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swint_under_rocprof.log 2>&1 || exit 1
cp $(find /tmp/prof_swint -name "*kernel_stats.csv" | head -1) $O/swint_rocprofv3_kernel_stats.csv
python3 tools/summarize_trace.py $(find /tmp/prof_swint -name "*kernel_trace.csv" | head -1) $O/swint_steady_state_per_step.csv 2 k_scatter_col 4 > $O/swint_steady_summary.txt; cat $O/swint_steady_summary.txt
echo "[3] Swin-T done"
timeout -k 10 500 python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_line_swinb_roberta_fp16.json 2> $O/bench_swinb.err || { tail -5 $O/bench_swinb.err; exit 1; }
rm -rf /tmp/prof_swinb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swinb -- python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swinb_under_rocprof.log 2>&1 || exit 1
cp $(find /tmp/prof_swinb -name "*kernel_stats.csv" | head -1) $O/swinb_roberta_fp16_rocprofv3_kernel_stats.csv
python3 tools/summarize_trace.py $(find /tmp/prof_swinb -name "*kernel_trace.csv" | head -1) $O/swinb_roberta_fp16_steady_state_per_step.csv 2 k_scatter_col 4 > $O/swinb_steady_summary.txt; cat $O/swinb_steady_summary.txt
echo "[4] Swin-B + RoBERTa fp16 done"
timeout -k 10 200 python3 tools/bench_lfm_dft.py 2>&1 | grep -v amdgpu.ids > $O/lfm_dft.txt
timeout -k 10 200 python3 tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids > $O/conv3x3_wgrad.txt
timeout -k 10 200 python3 tools/bench_r4_convs.py 2>&1 | grep -v amdgpu.ids > $O/remaining_miopen_convs.txt
BACKBONE=resnet101 TOP=120 timeout -k 10 400 python3 tools/glue_census.py > $O/glue_census_by_module.txt 2>&1
echo "[5] micro-benchmarks + census done"
