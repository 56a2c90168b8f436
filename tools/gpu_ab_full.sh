#!/bin/bash
# a FULL build kept under build/dev/<name>.so (cp pvtrace_amd/csrc/libpvtrace_hip.so there before a change) against the in-tree
# library, alternating, on the bench's configs: median fenced window, sustained, lone launches
# usage: tools/gpu_ab_full.sh name cfg...
mkdir -p gpurun_out
old=$1; shift
for r in 1 2; do
  echo "== in-tree"; bash tools/gpu_configs_quick.sh tree "$@"
  echo "== $old";    PVT_LIB=$PWD/build/dev/$old.so bash tools/gpu_configs_quick.sh $old "$@"
done
echo "== lone launches in-tree"; python tools/gpu_lone_launch.py 2>&1 | grep -v amdgpu
echo "== lone launches $old"; PVT_LIB=$PWD/build/dev/$old.so python tools/gpu_lone_launch.py 2>&1 | grep -v amdgpu
