#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
rm -rf /tmp/prof_swinb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_swinb -- python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/swinb_under_rocprof.log 2>&1 || { tail -5 gpurun_out/r4/swinb_under_rocprof.log; exit 1; }
F=$(find /tmp/prof_swinb -name "*kernel_trace.csv" | head -1)
python3 tools/summarize_trace.py $F gpurun_out/r4/swinb_steady.csv 2 k_scatter_col 4
cp $(find /tmp/prof_swinb -name "*kernel_stats.csv" | head -1) gpurun_out/r4/swinb_kernel_stats.csv
head -30 gpurun_out/r4/swinb_steady.csv | cut -c1-90,150-230
