#!/bin/bash
F="--no-cpu-baseline --repeats 6 --sustained-s 2.0 --total-photons 0 --extra-configs none"
run() { label=$1; shift
  env "$@" timeout 600 python bench.py $F $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: value %.4e  sustained %.4e  kernel_ms %.3f' % (d['value'], d['sustained']['value'], d['roofline']['kernel_ms_mean']))
"; }
for rep in 1 2; do for v in ws1 ws0; do
  run "$v" PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$v.so
done; done
python tools/gpu_history.py 2>&1 | tail -3
