#!/bin/bash
# the GPU test suite from a given test file on (after a fix), logs into gpurun_out/r4/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
tag=${1:-a}
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r4/full_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/full_gpu_$tag.log
tail -6 gpurun_out/r4/full_gpu_$tag.log
