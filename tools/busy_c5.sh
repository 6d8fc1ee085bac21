#!/bin/bash
# GPU-busy time per step of BASELINE config #5 under rocprofv3 for each environment setting given (wall-clock A/Bs of this config are
# noisy by +-3 ms: the eager text encoder and the host side of the scaler sit in the step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  export $v
  rm -rf /tmp/prof_c5
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -- python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/ab/c5_prof.log 2>&1 || { tail -3 gpurun_out/ab/c5_prof.log; exit 1; }
  echo "$v: $(python3 tools/summarize_trace.py $(find /tmp/prof_c5 -name '*kernel_trace.csv' | head -1) gpurun_out/ab/c5_steady.csv 2 k_scatter_col 4)"
done
