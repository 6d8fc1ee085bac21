#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_msda_gpu.py -x -q > gpurun_out/r4/test_msda.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/test_msda.log
tail -5 gpurun_out/r4/test_msda.log
GV_PATHS=0 timeout -k 10 300 python3 tools/bench_msda_gv.py > gpurun_out/r4/gv_col3.log 2>&1
OCPG_MSDA_COL_LP=2 GV_PATHS=0 timeout -k 10 300 python3 tools/bench_msda_gv.py > gpurun_out/r4/gv_col2.log 2>&1
cat gpurun_out/r4/gv_col3.log gpurun_out/r4/gv_col2.log
