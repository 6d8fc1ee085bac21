#!/bin/bash
# steady-state kernel summary of the B = 1 configuration (1 clip per step) next to the default's
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
rm -rf /tmp/prof_b1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b1 -- python3 bench.py --clips-per-gpu 1 --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-b1 > gpurun_out/ab/b1_prof.log 2>&1 || { tail -3 gpurun_out/ab/b1_prof.log; exit 1; }
python3 tools/summarize_trace.py $(find /tmp/prof_b1 -name '*kernel_trace.csv' | head -1) gpurun_out/ab/b1_steady.csv 3 k_scatter_col 4
