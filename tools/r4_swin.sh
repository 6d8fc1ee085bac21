#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 500 python3 bench.py --backbone video_swin_b_p4w7 --dtype fp16 --text roberta --frames 8 --height 480 --width 854 --clips-per-gpu 1 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench_swinb.json 2> gpurun_out/r4/bench_swinb.err; echo "swinb rc=$?"; tail -2 gpurun_out/r4/bench_swinb.err | cut -c1-300
timeout -k 10 400 python3 bench.py --backbone video_swin_t_p4w7 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench_swint.json 2> gpurun_out/r4/bench_swint.err; echo "swint rc=$?"; tail -2 gpurun_out/r4/bench_swint.err | cut -c1-300
python3 - <<'PY'
import json
for n in ("swinb","swint"):
    try:
        l=json.load(open(f"gpurun_out/r4/bench_{n}.json")); print(n, {k:l[k] for k in ("value","ms_per_step")}, l["config"]["launch"])
    except Exception as e: print(n, "no line", e)
PY
