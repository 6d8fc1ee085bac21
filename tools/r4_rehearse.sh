#!/bin/bash
# one-GPU rehearsal of the N = 2 launch path (gloo, both ranks on cuda:0): spawn, GEMM plan broadcast, launch-mode ladder, three graphs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
OCPG_REHEARSE_ONE_GPU=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/rehearse2.json 2> gpurun_out/r4/rehearse2.err; echo "rc=$?"
tail -3 gpurun_out/r4/rehearse2.err
python3 -c "
import json; l=json.loads([x for x in open('gpurun_out/r4/rehearse2.json').read().splitlines() if x.startswith('{')][-1]); print({k: l.get(k) for k in ('n_gpus','ms_per_step','ranks','hipgraph')}); print(l['config'])"
