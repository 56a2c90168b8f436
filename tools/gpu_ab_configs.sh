#!/bin/bash
# Developer tool (GPU box): like gpu_ab.sh, with the cfg4 / cfg5 legs -- fenced 10^7 windows (median, min, max) and
# the sustained rate of each.   usage: tools/gpu_ab_configs.sh name [name ...]
mkdir -p gpurun_out
one() {
  python bench.py 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s cfg2 %.4e sust %.4e |' % ('$1', d['value'], d['sustained']['value']), ' | '.join('%s %.3e [%.3e %.3e] sust %.3e' % (k, v['value'], v['min'], v['max'], v['sustained']['value']) for k, v in d['configs'].items()))"
}
for r in $(seq ${AB_ROUNDS:-2}); do
  one tree
  for v in "$@"; do PVT_LIB=$PWD/build/dev/$v.so one $v; done
done
