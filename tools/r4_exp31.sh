#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for v in 1 0 1 0; do
OCPG_GATE_BF=$v timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_bf$v.json 2> gpurun_out/r4/bench_bf$v.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_bf$v.json')); print('gate_bf=$v', l['ms_per_step'], l['value'])"
done
