#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/ab_env.sh "OCPG_X=1" "OCPG_X=0" ...  -- the default bench command once per environment
# setting, back to back on the same box (the A/B protocol behind the "same-box" numbers of DESIGN section 5); prints ms/step and clips/s.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  env $v timeout -k 10 500 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/ab/line.json 2> gpurun_out/ab/line.err || { tail -5 gpurun_out/ab/line.err; exit 1; }
  python3 -c "
import json; l=json.loads([x for x in open('gpurun_out/ab/line.json').read().splitlines() if x.startswith('{')][-1]); print('$v', round(l['ms_per_step'], 3), round(l['value'], 2))"
done
