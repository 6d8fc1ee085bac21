#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for v in 0 1; do
OCPG_GEMM_TUNE_FP32=$v timeout -k 10 500 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_tf$v.json 2> gpurun_out/r4/bench_tf$v.err || { tail -5 gpurun_out/r4/bench_tf$v.err; exit 1; }
python3 -c "
import json,sys; l=json.loads([x for x in open('gpurun_out/r4/bench_tf$v.json').read().splitlines() if x.startswith('{')][-1]); print('tune_fp32=$v', l['ms_per_step'], l['value'], l.get('gemm_plans'), l.get('final_loss'))"
done
