#!/bin/bash
# Developer tool (GPU box): the bench's cfg2 loop alternating between dev builds of THIS tree (build/dev/<name>.so,
# tools/dev_build.sh), same box.  usage: tools/gpu_ab_libs.sh "a.so b.so" [rounds] [extra bench flags]
libs=$1; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do
  for lib in $libs; do
    PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$lib python bench.py --extra-configs none --no-cpu-baseline --total-photons 0 --repeats 7 --sustained-s 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'median %.4e sustained %.4e' % (d['value'], d['sustained']['value']))"
  done
done
