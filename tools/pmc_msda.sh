#!/bin/bash
# rocprofv3 PMC passes (one counter set per pass, kernel-trace only) on the MSDA micro-benchmark at N=10 frames;
# writes gpurun_out/pmc_{f,w,a}.csv (counter_collection rows of the msda kernels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MSDA_FRAMES=10
for spec in "f FETCH_SIZE" "w WRITE_SIZE" "a TCC_EA0_ATOMIC_sum"; do
  set -- $spec
  rm -rf /tmp/pmc_$1
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -- python tools/bench_msda.py > /dev/null 2>&1 || exit 1
  F=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
  head -1 $F > gpurun_out/pmc_$1.csv
  grep -E "msda_bwd_tiled|msda_fwd_fast" $F >> gpurun_out/pmc_$1.csv
done
