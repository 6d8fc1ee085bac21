#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
OCPG_STEP_PHASES=1 timeout -k 10 500 python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_swinb_ph.json 2> gpurun_out/r4/bench_swinb_ph.err; echo "rc=$?"
python3 -c "
import json; l=json.load(open('gpurun_out/r4/bench_swinb_ph.json')); print(l['ms_per_step'], l.get('step_phases_ms'))"
