#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
GV_SELECT=1 timeout -k 10 300 python3 tools/bench_msda_gv.py 2>&1 | grep "selected" > gpurun_out/r4/gv_select2.log; cat gpurun_out/r4/gv_select2.log
echo "--- start on tiled, never return" 
OCPG_MSDA_SEL_TO_COL=-1 GV_SELECT_START=1 GV_SELECT=1 timeout -k 10 300 python3 tools/bench_msda_gv.py 2>&1 | grep "selected"
