#!/bin/bash
# usage: tools/ab_variants.sh "<variant names ('' = default lib)>" [ENV=VAL ...] -- bench_msda.py under each library build
vars=$1; shift
for kv in "$@"; do export "$kv"; done
for v in $vars; do
  lib=$PWD/ocpg_amd/lib/libocpg_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/ocpg_amd/lib/libocpg_hip.so
  OCPG_HIP_LIB=$lib timeout -k 10 120 python tools/bench_msda.py 2>&1 | grep "Lq=\|col vs" | sed "s/^/$v /"
done
