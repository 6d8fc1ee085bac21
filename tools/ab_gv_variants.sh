#!/bin/bash
# usage: tools/ab_gv_variants.sh "<variant names ('default' = the shipped library)>" [ENV=VAL ...]
# grad_value micro-benchmark (tools/bench_msda_gv.py, column path, ring offsets) under each library build, warm and with cold caches
vars=$1; shift
for kv in "$@"; do export "$kv"; done
for v in $vars; do
  lib=$PWD/ocpg_amd/lib/libocpg_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/ocpg_amd/lib/libocpg_hip.so
  for cold in 0 1; do
    OCPG_HIP_LIB=$lib GV_COLD=$cold GV_MODES=${GV_MODES:-ring} GV_PATHS=0 GV_NOCHECK=1 timeout -k 10 120 python tools/bench_msda_gv.py 2>&1 | grep '"us"' | sed "s/^/$v cold=$cold /"
  done
done
