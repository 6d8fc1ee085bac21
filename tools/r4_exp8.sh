#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_msda_gpu.py -x -q > gpurun_out/r4/test_msda_c4b.log 2>&1; tail -3 gpurun_out/r4/test_msda_c4b.log
for lp in 4 3 2; do
echo "LP=$lp" >> gpurun_out/r4/gv_margin5.log
OCPG_MSDA_COL_LP=$lp GV_PATHS=0 timeout -k 10 300 python3 tools/bench_msda_gv.py >> gpurun_out/r4/gv_margin5.log 2>&1
done
echo "LP=4 margin 4" >> gpurun_out/r4/gv_margin5.log
OCPG_MSDA_MARGIN_LO=4 GV_PATHS=0 GV_MODES=ring timeout -k 10 300 python3 tools/bench_msda_gv.py >> gpurun_out/r4/gv_margin5.log 2>&1
grep -v amdgpu gpurun_out/r4/gv_margin5.log
