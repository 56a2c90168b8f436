#!/bin/bash
# claim size sweep (dev builds), carry on/off, 1e6 and 4e6 photons per launch
F="--no-cpu-baseline --repeats 2 --sustained-s 0.7 --total-photons 0 --extra-configs none"
run() { label=$1; shift
  env "$@" timeout 600 python bench.py $F $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: value %.3e  sustained %.3e  kernel_ms %.3f' % (d['value'], d['sustained']['value'], d['roofline']['kernel_ms_mean']))
"; }
for n in 1000000 4000000; do for c in 1 2 4; do
  EXTRA="--photons $n --ray-buffers 3"
  run "claim $c n $n nocarry s3w2" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/claim$c.so PVT_NO_CARRY=1
  run "claim $c n $n carry   s3w1" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/claim$c.so PVT_PIPE_WGS=1
  run "claim $c n $n carry   s3w2" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/claim$c.so
  EXTRA="--photons $n --ray-buffers 3 --streams 1"
  run "claim $c n $n carry   s1w4" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/claim$c.so
done; done
