#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
GV_PATHS=0 timeout -k 10 300 python3 tools/bench_msda_gv.py > gpurun_out/r4/gv_col3b.log 2>&1
OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_stamps.so GV_MODES=ring timeout -k 10 300 python3 tools/stamps_col3.py > gpurun_out/r4/stamps_col3b_ring.log 2>&1
cat gpurun_out/r4/gv_col3b.log gpurun_out/r4/stamps_col3b_ring.log
timeout -k 10 600 python3 -m pytest tests/test_msda_gpu.py -x -q -k "self_attention" > gpurun_out/r4/test_msda_b.log 2>&1; tail -3 gpurun_out/r4/test_msda_b.log
