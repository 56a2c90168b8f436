#!/bin/bash
# the lone-wave step (tools/gpu_lone_step.py) of the in-tree library and of dev builds build/dev/<name>.so, same box
# usage: tools/gpu_lone_ab.sh name...
mkdir -p gpurun_out
{
echo "== in-tree"; timeout 300 python tools/gpu_lone_step.py 2>&1 | grep "per step"
for n in "$@"; do
  echo "== $n"; PVT_LIB=$PWD/build/dev/$n.so timeout 300 python tools/gpu_lone_step.py 2>&1 | grep "per step"
done
} | tee gpurun_out/lone_ab.txt
