#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
echo "== full"; timeout -k 10 200 python3 tools/bench_lfm_dft.py 2>&1 | grep -v amdgpu.ids | head -2
echo "== no compute (memory passes only)"; OCPG_HIP_LIB=$PWD/ocpg_amd/lib/libocpg_hip_nocompute.so timeout -k 10 200 python3 tools/bench_lfm_dft.py 2>&1 | grep -v amdgpu.ids | head -2
