#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
bash tools/pmc_gv.sh > gpurun_out/r4/pmc_gv.log 2>&1; tail -30 gpurun_out/r4/pmc_gv.log
timeout -k 10 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4/bench_b.json 2> gpurun_out/r4/bench_b.err; echo "bench rc=$?"; tail -3 gpurun_out/r4/bench_b.err
python3 - <<PY
import json
l=json.load(open("gpurun_out/r4/bench_b.json"))
print({k:l[k] for k in ("value","ms_per_step")}, l.get("b1"), "\n", l.get("roofline"), "\n", l.get("roofline_ring_offsets"), "\n", l["hipgraph"], l["config"]["launch"])
PY
