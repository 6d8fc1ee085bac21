#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py -x -q -k "splitk or e2e or lfm or conv3x3" > gpurun_out/r4/test_splitk.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/test_splitk.log
tail -5 gpurun_out/r4/test_splitk.log
for f in 1 0; do
OCPG_SPLITK_3X3=$f timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-b1 --no-kernel-timing > gpurun_out/r4/bench_sk$f.json 2> gpurun_out/r4/bench_sk$f.err; echo "bench rc=$?"
python3 - <<PY
import json
l=json.load(open("gpurun_out/r4/bench_sk$f.json"))
print("splitk=$f", {k:l[k] for k in ("value","ms_per_step")})
PY
done
