#!/bin/bash
# one-GPU rehearsal of the N = 2 launch path (gloo, both ranks on cuda:0) for the configurations that joined the three-graph step in the
# second half of round 4: Video-Swin-T bf16 (config #4) and Video-Swin-T fp16 + GradScaler (config #5's precision on the small backbone);
# never a measurement -- it shows that spawn, plan broadcast, the ladder's first rung and the bucketed reduce path run end to end
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for dt in bf16 fp16; do
  OCPG_REHEARSE_ONE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --backbone video_swin_t_p4w7 --dtype $dt --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/r4/rehearse2_swint_$dt.json 2> gpurun_out/r4/rehearse2_swint_$dt.err; echo "$dt rc=$?"
  tail -2 gpurun_out/r4/rehearse2_swint_$dt.err | cut -c1-300
  python3 -c "
import json; l=json.loads([x for x in open('gpurun_out/r4/rehearse2_swint_$dt.json').read().splitlines() if x.startswith('{')][-1]); print({k: l.get(k) for k in ('n_gpus','dtype','ms_per_step','ranks','hipgraph','final_loss')}); print(l['config']['launch'], l['config'].get('launch_fallback'))"
done
