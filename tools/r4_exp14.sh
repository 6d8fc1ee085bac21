#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_msda_gpu.py -x -q -k "selection or fused or self_attention" > gpurun_out/r4/test_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/test_sel.log
tail -5 gpurun_out/r4/test_sel.log
GV_SELECT=1 timeout -k 10 300 python3 tools/bench_msda_gv.py 2>&1 | grep -v amdgpu | grep "selected\|column"
