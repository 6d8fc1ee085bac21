#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 256 512; do
echo "== two sets, occupancy 1, WGs $w"; OCPG_WGRAD_WGS=$w timeout -k 10 100 python3 tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids
echo "== one set, occupancy 2, WGs $w"; OCPG_WGRAD_WGS=$w OCPG_HIP_LIB=$PWD/ocpg_amd/lib/libocpg_hip_wg1.so timeout -k 10 100 python3 tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids
done
