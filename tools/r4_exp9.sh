#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_msda_gpu.py -x -q -k "self_attention" > gpurun_out/r4/test_msda_c4c.log 2>&1; tail -3 gpurun_out/r4/test_msda_c4c.log
GV_PATHS=0 timeout -k 10 300 python3 tools/bench_msda_gv.py 2>&1 | grep -v amdgpu
