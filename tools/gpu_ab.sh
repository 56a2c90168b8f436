#!/bin/bash
# Developer tool (GPU box): A/B of builds of the library on the same box, alternating -- the in-tree library against
# build/dev/<name>.so (PVT_LIB) for every name given; prints the bench's median, sustained and strong legs of each run.
# usage: tools/gpu_ab.sh name [name ...]      (AB_ROUNDS=n rounds, default 2)
mkdir -p gpurun_out
one() {
  python bench.py --extra-configs none 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s value %.4e sustained %.4e strong %.4e kernel_ms %.3f' % ('$1', d['value'], d['sustained']['value'], d['strong_scaling']['value'], d['roofline']['kernel_ms_mean']))"
}
for r in $(seq ${AB_ROUNDS:-2}); do
  one tree
  for v in "$@"; do PVT_LIB=$PWD/build/dev/$v.so one $v; done
done
