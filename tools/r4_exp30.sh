#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "text_gate or lfm or groupnorm or e2e or full_size or fused_front" 2>&1 | tail -40 > gpurun_out/r4/t30.log; tail -5 gpurun_out/r4/t30.log
timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_bf.json 2> gpurun_out/r4/bench_bf.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_bf.json')); print('batch-first gate', l['ms_per_step'], l['value'])"
