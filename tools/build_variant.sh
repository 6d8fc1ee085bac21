#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]  -> ocpg_amd/lib/libocpg_hip_<name>.so (experiment builds; load with OCPG_HIP_LIB)
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  -o ocpg_amd/lib/libocpg_hip_$name.so ocpg_amd/csrc/*.hip -L/opt/rocm/lib -lhipblaslt
