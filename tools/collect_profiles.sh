#!/bin/bash
# Round-end measurement set (one gpurun call): bench lines (graph default / eager / 1 clip), rocprofv3 kernel trace of the
# default command, PMC passes of the MSDA kernels, Swin-T (config #4) and Swin-B + RoBERTa fp16 (config #5) runs.
# Everything lands in gpurun_out/collect/; copy what is to be judged into profiles/.
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
timeout -k 10 400 bash tools/pmc_msda.sh > $O/pmc_msda.txt 2>&1 || exit 1
cp gpurun_out/msda_pmc.json gpurun_out/pmc_f.csv gpurun_out/pmc_w.csv gpurun_out/pmc_a.csv $O/
echo "[4] PMC done"
timeout -k 10 400 python bench.py --backbone video_swin_t_p4w7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_line_swint.json 2> $O/bench_swint.err || { tail -5 $O/bench_swint.err; exit 1; }
rm -rf /tmp/prof_swint
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swint -- python bench.py --backbone video_swin_t_p4w7 --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swint_under_rocprof.log 2>&1 || exit 1
cp $(find /tmp/prof_swint -name "*kernel_stats.csv" | head -1) $O/swint_rocprofv3_kernel_stats.csv
python tools/summarize_trace.py $(find /tmp/prof_swint -name "*kernel_trace.csv" | head -1) $O/swint_steady_state_per_step.csv 2 k_scatter_col 4 > $O/swint_steady_summary.txt; cat $O/swint_steady_summary.txt
echo "[5] Swin-T done"
[ -n "$SKIP_SWINB" ] && exit 0
timeout -k 10 500 python bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_line_swinb_roberta_fp16.json 2> $O/bench_swinb.err || { tail -5 $O/bench_swinb.err; exit 1; }
rm -rf /tmp/prof_swinb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swinb -- python bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swinb_under_rocprof.log 2>&1 || exit 1
cp $(find /tmp/prof_swinb -name "*kernel_stats.csv" | head -1) $O/swinb_roberta_fp16_rocprofv3_kernel_stats.csv
python tools/summarize_trace.py $(find /tmp/prof_swinb -name "*kernel_trace.csv" | head -1) $O/swinb_steady_state_per_step.csv 2 k_scatter_col 4 > $O/swinb_steady_summary.txt; cat $O/swinb_steady_summary.txt
echo "[6] Swin-B + RoBERTa fp16 done"
