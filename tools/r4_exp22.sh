#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 300 python3 tools/bench_r4_convs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/convs.txt &&
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "swin_window or backbone or resnet or e2e or lfm or full_size or bottleneck" 2>&1 | tail -8
