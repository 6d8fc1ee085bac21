#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for v in 1 0; do
OCPG_LFM_DFT=$v timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_dft$v.json 2> gpurun_out/r4/bench_dft$v.err || exit 1
python3 -c "
import json,sys; l=json.load(open('gpurun_out/r4/bench_dft$v.json')); print('dft=$v', l['ms_per_step'], l['value'])"
done
