#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_msda_gpu.py -x -q -k "self_attention" > gpurun_out/r4/test_msda_c4.log 2>&1; tail -3 gpurun_out/r4/test_msda_c4.log
GV_PATHS=0 timeout -k 10 300 python3 tools/bench_msda_gv.py > gpurun_out/r4/gv_col4.log 2>&1; cat gpurun_out/r4/gv_col4.log
rm -f gpurun_out/r4/gv_c4cuts.log
for c in 2 3 4; do
  echo "cut $c" >> gpurun_out/r4/gv_c4cuts.log
  OCPG_HIP_LIB=$GRAFT_REPO_ROOT/ocpg_amd/lib/libocpg_hip_c4cut$c.so GV_NOCHECK=1 GV_PATHS=0 GV_MODES=ring timeout -k 10 300 python3 tools/bench_msda_gv.py >> gpurun_out/r4/gv_c4cuts.log 2>&1
done
cat gpurun_out/r4/gv_c4cuts.log
