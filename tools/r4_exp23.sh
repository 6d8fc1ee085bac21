#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "swin_window or backbone or resnet or e2e or lfm or full_size or bottleneck" 2>&1 | tail -30 > gpurun_out/r4/t23.log; tail -5 gpurun_out/r4/t23.log
for v in "1 64" "0 64" "1 128" "0 128"; do set -- $v
OCPG_STRIDED_1X1=$1 OCPG_MFMA_CONV3X3_MIN_C=$2 timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_s$1_c$2.json 2> gpurun_out/r4/bench_s$1_c$2.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_s$1_c$2.json')); print('strided=$1 minC=$2', l['ms_per_step'], l['value'])"
done
