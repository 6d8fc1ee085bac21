#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lfm or groupnorm or group_norm or e2e or full_size or fused_front" 2>&1 | tail -40 > gpurun_out/r4/t29.log; tail -5 gpurun_out/r4/t29.log
for v in "1 1" "0 1" "0 0"; do set -- $v
OCPG_GN_CL_OUT=$1 OCPG_CL_FUSE=$2 timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_cl$1$2.json 2> gpurun_out/r4/bench_cl$1$2.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_cl$1$2.json')); print('gn_cl=$1 cl_fuse=$2', l['ms_per_step'], l['value'])"
done
