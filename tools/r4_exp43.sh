#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for v in "OCPG_SPLIT_ROWS=1536" "OCPG_SPLIT_ROWS=768" "OCPG_SPLIT_ROWS=3072" "OCPG_SPLIT_ROWS=6144" "OCPG_SPLIT_K=0"; do
env $v timeout -k 10 500 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/bench_x.json 2> gpurun_out/r4/bench_x.err || { tail -5 gpurun_out/r4/bench_x.err; exit 1; }
python3 -c "
import json,sys; l=json.loads([x for x in open('gpurun_out/r4/bench_x.json').read().splitlines() if x.startswith('{')][-1]); print('$v', l['ms_per_step'], l['value'])"
done
