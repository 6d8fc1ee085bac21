#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/gv_c4cuts2.log
for c in cut9 cut5; do
  echo "$c" >> gpurun_out/r4/gv_c4cuts2.log
  OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_c4$c.so GV_NOCHECK=1 GV_PATHS=0 GV_MODES=ring timeout -k 10 300 python3 tools/bench_msda_gv.py >> gpurun_out/r4/gv_c4cuts2.log 2>&1
done
cat gpurun_out/r4/gv_c4cuts2.log
