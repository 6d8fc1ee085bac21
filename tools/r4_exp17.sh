#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -k "premasked or bottleneck or splitk or conv3x3_mfma or clip_adamw or fused_cast or amp_cache" > gpurun_out/r4/test_premask.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/test_premask.log
tail -5 gpurun_out/r4/test_premask.log
for f in 1 0; do
OCPG_PREMASK_DGRAD=$f timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-b1 --no-kernel-timing > gpurun_out/r4/bench_pm$f.json 2> gpurun_out/r4/bench_pm$f.err; echo "bench rc=$?"
python3 - <<PY
import json
l=json.load(open("gpurun_out/r4/bench_pm$f.json"))
print("premask=$f", {k:l[k] for k in ("value","ms_per_step")})
PY
done
