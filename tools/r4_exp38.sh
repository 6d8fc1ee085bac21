#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== full"; timeout -k 10 100 python3 tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids
echo "== no MFMA loop"; OCPG_HIP_LIB=$PWD/ocpg_amd/lib/libocpg_hip_wgnc.so timeout -k 10 100 python3 tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids
echo "== no park / fetch"; OCPG_HIP_LIB=$PWD/ocpg_amd/lib/libocpg_hip_wgnl.so timeout -k 10 100 python3 tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids
