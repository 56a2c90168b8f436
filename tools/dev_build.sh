#!/bin/bash
# Developer tool: fast build of the two analytic-scene variants only (tally + history) into build/dev/<name>.so
# usage: tools/dev_build.sh name [extra hipcc flags]      (DEVV=2: the mesh variants instead)
name=$1; shift
mkdir -p /root/repo/build/dev
cd /root/repo/pvtrace_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -mllvm -disable-machine-licm -fno-unroll-loops -fPIC -shared -DPVT_DEV_VARIANTS=${DEVV:-1} "$@" pvt_trace.hip -o /root/repo/build/dev/$name.so
