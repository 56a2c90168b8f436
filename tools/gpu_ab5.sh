#!/bin/bash
F="--no-cpu-baseline --repeats 4 --sustained-s 1.0 --total-photons 0 --extra-configs none"
run() { label=$1; shift
  env "$@" timeout 600 python bench.py $F $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: value %.3e  sustained %.3e  kernel_ms %.3f' % (d['value'], d['sustained']['value'], d['roofline']['kernel_ms_mean']))
"; }
for c in 16 32 64; do
  EXTRA="--streams 3"; run "end$c s3w2" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/end$c.so
  EXTRA="--streams 2"; run "end$c s2w2" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/end$c.so PVT_PIPE_WGS=2
  EXTRA="--streams 1"; run "end$c s1w4" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/end$c.so
  EXTRA="--streams 3"; run "end$c s3w2 nocarry" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/end$c.so PVT_NO_CARRY=1
done
