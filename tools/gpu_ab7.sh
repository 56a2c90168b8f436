#!/bin/bash
F="--no-cpu-baseline --repeats 10 --sustained-s 0 --total-photons 0 --extra-configs none"
run() { label=$1; shift
  env "$@" timeout 600 python bench.py $F $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: value %.4e  min %.4e max %.4e kernel_ms %.3f' % (d['value'], d['repeats']['min'], d['repeats']['max'], d['roofline']['kernel_ms_mean']))
"; }
for rep in 1 2; do
EXTRA=""; run "tail_wide=1" PVT_TAIL_WIDE=1
EXTRA=""; run "tail_wide=0" PVT_TAIL_WIDE=0
EXTRA=""; run "tail_wide=2" PVT_TAIL_WIDE=2
EXTRA=""; run "tail_wide=3" PVT_TAIL_WIDE=3
done
