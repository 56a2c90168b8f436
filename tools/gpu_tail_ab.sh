#!/bin/bash
# a dev build with a change to the drain tail against the in-tree library, same box: parity of the scenes the dev build
# serves, the lone-wave step, lone launches of BASELINE's sizes, then the bench's stream (the headline must not move)
# usage: tools/gpu_tail_ab.sh name...     (build/dev/<name>.so)
mkdir -p gpurun_out
{
for n in "$@"; do
  echo "== parity with $n"
  PVT_LIB=$PWD/build/dev/$n.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_carry.py tests/test_gpu_records.py -q -x \
     -k "lsc_equivalent or nested_cylinders or bench_slab or hello_world or fresnel_box or touching or trapped or kitchen or cfg2 or coated or hist_slab or lambertian" 2>&1 | tail -3
done
echo "== lone step"; timeout 300 python tools/gpu_lone_step.py 2>&1 | grep "per step"
for n in "$@"; do echo "== lone step $n"; PVT_LIB=$PWD/build/dev/$n.so timeout 300 python tools/gpu_lone_step.py 2>&1 | grep "per step"; done
echo "== lone launches"; timeout 300 python tools/gpu_lone_launch.py 2>&1 | grep -v amdgpu.ids
for n in "$@"; do echo "== lone launches $n"; PVT_LIB=$PWD/build/dev/$n.so timeout 300 python tools/gpu_lone_launch.py cfg2 2>&1 | grep -v amdgpu.ids; done
AB_ROUNDS=${AB_ROUNDS:-2} bash tools/gpu_ab.sh "$@"
} 2>&1 | tee gpurun_out/tail_ab.txt
