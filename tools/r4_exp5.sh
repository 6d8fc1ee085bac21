#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for c in 1 2 3 4; do
  echo "cut $c" >> gpurun_out/r4/gv_cuts.log
  OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_cut$c.so GV_NOCHECK=1 GV_PATHS=0 GV_MODES=ring,trained timeout -k 10 300 python3 tools/bench_msda_gv.py >> gpurun_out/r4/gv_cuts.log 2>&1
done
cat gpurun_out/r4/gv_cuts.log
