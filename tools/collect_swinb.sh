#!/bin/bash
# Config #5 (Video-Swin-B + RoBERTa, fp16 + GradScaler, one 8 x 480 x 854 clip per step): bench line + steady-state rocprofv3 cut.
# Step [6] of tools/collect_profiles.sh as its own gpurun call (the full set does not fit one call's time limit).
O=gpurun_out/collect; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_line_swinb_roberta_fp16.json 2> $O/bench_swinb.err || { tail -5 $O/bench_swinb.err; exit 1; }
echo "[6a] bench line done"; cut -c1-400 $O/bench_line_swinb_roberta_fp16.json
rm -rf /tmp/prof_swinb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swinb -- python bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > $O/swinb_under_rocprof.log 2>&1 || { tail -5 $O/swinb_under_rocprof.log; exit 1; }
cp $(find /tmp/prof_swinb -name "*kernel_stats.csv" | head -1) $O/swinb_roberta_fp16_rocprofv3_kernel_stats.csv
python tools/summarize_trace.py $(find /tmp/prof_swinb -name "*kernel_trace.csv" | head -1) $O/swinb_steady_state_per_step.csv 2 k_scatter_col 4 > $O/swinb_steady_summary.txt; cat $O/swinb_steady_summary.txt
echo "[6] Swin-B + RoBERTa fp16 done"
