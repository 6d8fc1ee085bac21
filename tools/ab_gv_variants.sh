#!/bin/bash
# usage: tools/ab_gv_variants.sh "<variant names ('default' = the shipped library)>" [ENV=VAL ...]
# grad_value micro-benchmark (tools/bench_msda_gv.py, ring offsets unless GV_MODES says otherwise) under each library build
# (tools/build_variant_one.sh <name> ocpg_amd/csrc/msda_col.hip -D...), back to back (cold=0) and on cold operands (cold=1: GV_COLD=1).
#   default                         the column kernel alone (ocpg_msda_bwd_value_f32)
#   GV_SELECT=1 [GV_PATHS=]         + the SELECTING entry point (ocpg_msda_bwd_value_sel_f32: active kernel, the idle launches of the other
#                                   family, the commit); GV_PATHS= (empty) times only that
vars=$1; shift
for kv in "$@"; do export "$kv"; done
for v in $vars; do
  lib=$PWD/ocpg_amd/lib/libocpg_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/ocpg_amd/lib/libocpg_hip.so
  for cold in 0 1; do
    OCPG_HIP_LIB=$lib GV_COLD=$cold GV_MODES=${GV_MODES:-ring} GV_PATHS=${GV_PATHS-0} GV_NOCHECK=1 timeout -k 10 120 python tools/bench_msda_gv.py 2>&1 | grep '"us"' | sed "s/^/$v cold=$cold /"
  done
done
