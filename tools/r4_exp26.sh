#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lfm" 2>&1 | tail -40 > gpurun_out/r4/t24.log; tail -3 gpurun_out/r4/t24.log
timeout -k 10 200 python3 tools/bench_lfm_dft.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/lfm_dft.txt
