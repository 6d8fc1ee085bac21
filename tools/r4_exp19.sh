#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_win_attn_gpu.py tests/test_model_gpu.py -x -q -k "window or swin or fused_autocast" > gpurun_out/r4/test_win.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/test_win.log
tail -4 gpurun_out/r4/test_win.log
WIN_ATTN_MODES=1 bash tools/prof_any.sh r4_win tools/bench_win_attn.py 2>&1 | head -8
grep "mfma" gpurun_out/prof_r4_win.log
