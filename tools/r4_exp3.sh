#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_stamps.so GV_MODES=ring timeout -k 10 300 python3 tools/stamps_col3.py > gpurun_out/r4/stamps_col3_ring.log 2>&1
OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_stamps.so GV_MODES=trained timeout -k 10 300 python3 tools/stamps_col3.py > gpurun_out/r4/stamps_col3_trained.log 2>&1
cat gpurun_out/r4/stamps_col3_ring.log gpurun_out/r4/stamps_col3_trained.log
