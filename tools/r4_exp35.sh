#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
echo "== NT=256"; timeout -k 10 200 python3 tools/bench_lfm_dft.py 2>&1 | grep -v amdgpu.ids
echo "== NT=512"; OCPG_HIP_LIB=$PWD/ocpg_amd/lib/libocpg_hip_nt512.so timeout -k 10 200 python3 tools/bench_lfm_dft.py 2>&1 | grep -v amdgpu.ids
