#!/bin/bash
# the same A/B runner for BASELINE config #5 (Video-Swin-B + RoBERTa, fp16, 8 x 480 x 854, 1 clip)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  env $v timeout -k 10 500 python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/ab/line.json 2> gpurun_out/ab/line.err || { tail -5 gpurun_out/ab/line.err; exit 1; }
  python3 -c "
import json; l=json.loads([x for x in open('gpurun_out/ab/line.json').read().splitlines() if x.startswith('{')][-1]); print('$v', round(l['ms_per_step'], 3), round(l['value'], 2))"
done
