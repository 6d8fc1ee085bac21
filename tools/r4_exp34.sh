#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for v in "X=1" "OCPG_LFM_DFT=0" "OCPG_SMALL_LINEAR_F32=0" "OCPG_STRIDED_1X1=0" "OCPG_LS_FEAT_N16=0" "OCPG_GN_CL_OUT=0"; do
echo "== $v"
env $v timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "full_size_step_vs_oracle" 2>&1 | grep -E "referee|passed|failed|fp64" | head -8
done
