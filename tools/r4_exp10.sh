#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_msda_gpu.py -x -q -k "self_attention" > gpurun_out/r4/test_msda_c4d.log 2>&1; tail -3 gpurun_out/r4/test_msda_c4d.log
GV_PATHS=0 GV_MODES=ring timeout -k 10 300 python3 tools/bench_msda_gv.py 2>&1 | grep -v amdgpu
for h in 24 160 100000; do echo "heavy $h"; OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_hp$h.so GV_PATHS=0 GV_MODES=ring timeout -k 10 300 python3 tools/bench_msda_gv.py 2>&1 | grep -v amdgpu; done
