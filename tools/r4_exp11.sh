#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_msda_gpu.py tests/test_model_gpu.py -x -q -k "msda or transformer or e2e or selection or fused or train_step" > gpurun_out/r4/test_fused.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/test_fused.log
tail -5 gpurun_out/r4/test_fused.log
GV_SELECT=1 timeout -k 10 300 python3 tools/bench_msda_gv.py 2>&1 | grep -v amdgpu > gpurun_out/r4/gv_select.log; cat gpurun_out/r4/gv_select.log
for f in 1 0; do
OCPG_MSDA_FUSED_FRONT=$f timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-b1 > gpurun_out/r4/bench_fused$f.json 2> gpurun_out/r4/bench_fused$f.err; echo "bench rc=$?"
python3 - <<PY
import json
l=json.load(open("gpurun_out/r4/bench_fused$f.json"))
print("fused=$f", {k:l[k] for k in ("value","ms_per_step")}, l.get("roofline",{}).get("launch_us"))
PY
done
