#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
GV_MODES=ring ITERS=20 bash tools/prof_any.sh r4_gv_ring tools/bench_msda_gv.py > gpurun_out/r4/gv_ring_kernels.txt 2>&1
cp gpurun_out/prof_r4_gv_ring.log gpurun_out/r4/gv_ring.log
GV_MODES=trained ITERS=20 bash tools/prof_any.sh r4_gv_trained tools/bench_msda_gv.py > gpurun_out/r4/gv_trained_kernels.txt 2>&1
cp gpurun_out/prof_r4_gv_trained.log gpurun_out/r4/gv_trained.log
OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_noflush.so GV_PATHS=0 GV_NOCHECK=1 python3 tools/bench_msda_gv.py > gpurun_out/r4/gv_noflush.log 2>&1
cat gpurun_out/r4/gv_ring_kernels.txt gpurun_out/r4/gv_trained_kernels.txt gpurun_out/r4/gv_noflush.log
