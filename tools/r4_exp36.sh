#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "conv3x3_mfma_kernel or premasked or splitk or e2e" 2>&1 | tail -30 > gpurun_out/r4/t36.log; tail -4 gpurun_out/r4/t36.log
for v in 1 0; do
OCPG_DGRAD_OWN_WEIGHT=$v timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_dw$v.json 2> gpurun_out/r4/bench_dw$v.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_dw$v.json')); print('dgrad_own_weight=$v', l['ms_per_step'], l['value'])"
done
