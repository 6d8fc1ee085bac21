#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "conv3x3_mfma_kernel or conv3x3_own or premasked or e2e" 2>&1 | tail -30 > gpurun_out/r4/t37.log; tail -4 gpurun_out/r4/t37.log
for v in 1 0; do
OCPG_WGRAD_OWN=$v timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_wg$v.json 2> gpurun_out/r4/bench_wg$v.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_wg$v.json')); print('wgrad_own=$v', l['ms_per_step'], l['value'])"
done
