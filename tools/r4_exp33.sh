#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_msda_gpu.py -x -q -m gpu -k "small_linear or msda_module or decoder or e2e or full_size or deformable" 2>&1 | tail -40 > gpurun_out/r4/t33.log; tail -4 gpurun_out/r4/t33.log
for v in 1 0; do
OCPG_SMALL_LINEAR_F32=$v timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_sl$v.json 2> gpurun_out/r4/bench_sl$v.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_sl$v.json')); print('sl_f32=$v', l['ms_per_step'], l['value'])"
done
