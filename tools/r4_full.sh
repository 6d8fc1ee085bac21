#!/bin/bash
# full GPU test suite + the default bench line (round-4 checkpoints): logs into gpurun_out/r4/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
tag=${1:-a}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4/full_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/full_gpu_$tag.log
tail -4 gpurun_out/r4/full_gpu_$tag.log
timeout -k 10 600 python3 bench.py > gpurun_out/r4/bench_$tag.json 2> gpurun_out/r4/bench_$tag.err; echo "bench rc=$?"
python3 - <<PY
import json
l=json.load(open("gpurun_out/r4/bench_$tag.json"))
print({k:l[k] for k in ("value","ms_per_step")}, l.get("b1"), l.get("roofline"), l.get("roofline_trained_like_offsets"))
PY
