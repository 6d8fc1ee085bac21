#!/bin/bash
# usage: tools/ab_gv_sel.sh "<variant names>" [ENV=VAL ...] -- the SELECTING grad_value entry point (ocpg_msda_bwd_value_sel_f32: column kernel
# + the idle launches of the tiled family + the commit), ring offsets, warm and on cold operands, under each library build
vars=$1; shift
for kv in "$@"; do export "$kv"; done
for v in $vars; do
  lib=$PWD/ocpg_amd/lib/libocpg_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/ocpg_amd/lib/libocpg_hip.so
  for cold in 0 1; do
    OCPG_HIP_LIB=$lib GV_COLD=$cold GV_MODES=${GV_MODES:-ring} GV_PATHS=${GV_PATHS:-} GV_SELECT=1 GV_NOCHECK=1 timeout -k 10 120 python tools/bench_msda_gv.py 2>&1 | grep '"us"' | sed "s/^/$v cold=$cold /"
  done
done
