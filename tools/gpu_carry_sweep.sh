#!/bin/bash
# Developer tool (GPU box): carry tests, then per-launch timelines (tools/gpu_timeline2.sh) of the bench with and without carried photons
timeout 600 python -m pytest tests/test_gpu_carry.py tests/test_gpu_engine_api.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
bash tools/gpu_timeline2.sh 2>&1 | grep -v "^+" | awk '/=== /{c=0} {c++; if (c<=9) print}'
F="--no-cpu-baseline --repeats 2 --sustained-s 1.0 --total-photons 0 --extra-configs none"
run() { label=$1; shift
  env "$@" timeout 600 python bench.py $F $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: value %.3e  sustained %.3e  kernel_ms %.3f' % (d['value'], d['sustained']['value'], d['roofline']['kernel_ms_mean']))
"; }
for s in 1 2 3; do for w in 1 2 4; do
  EXTRA="--streams $s"
  run "carry   s$s w$w" PVT_PIPE_WGS=$w
  run "nocarry s$s w$w" PVT_PIPE_WGS=$w PVT_NO_CARRY=1
done; done
