#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python3 tools/glue_census.py > gpurun_out/r4/glue_census.txt 2>&1; echo rc=$?
grep -n "per (module, op)" gpurun_out/r4/glue_census.txt | head -2
