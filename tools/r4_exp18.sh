#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py -x -q -k "deferred or bench_mode or whole_step or segmented" > gpurun_out/r4/test_defer.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/test_defer.log
tail -5 gpurun_out/r4/test_defer.log
bash tools/r4_prof.sh c 2>&1 | grep -E "ms/step|multi_cast|bn_act_bwd|window wall"
