#!/bin/bash
# usage: tools/build_variant_one.sh <name> <file.hip> [-DFLAG ...] -> ocpg_amd/lib/libocpg_hip_<name>.so: ONE source recompiled with the flags,
# every other object taken from the incremental build (ocpg_amd/lib/obj) -- seconds instead of minutes per experiment build
name=$1; src=$2; shift 2
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -std=c++17 -fPIC -Wno-unused-function "$@" -c $src -o /tmp/${base}_$name.o || exit 1
objs=$(ls ocpg_amd/lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ocpg_amd/lib/libocpg_hip_$name.so $objs /tmp/${base}_$name.o -L/opt/rocm/lib -lhipblaslt
