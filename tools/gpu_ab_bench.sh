#!/bin/bash
# Developer tool (GPU box): A/B of builds through bench.py, alternating.  usage: gpu_ab_bench.sh rounds lib1 lib2 ...
N=$1; shift
for i in $(seq $N); do
  for lib in "$@"; do
    PVT_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --total-photons 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$lib', 'headline %.3fe9 median %.3fe9 sustained %.3fe9' % (d['value']/1e9, d['repeats']['median']/1e9, d['sustained']['value']/1e9))"
  done
done
