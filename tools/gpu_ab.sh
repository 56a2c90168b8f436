#!/bin/bash
# A/B of the bench under developer switches. usage: tools/gpu_ab.sh
mkdir -p gpurun_out
F="--no-cpu-baseline --repeats 4 --sustained-s 1.0 --total-photons 0"
timeout 600 python -m pytest tests/test_gpu_carry.py tests/test_gpu_engine_api.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
for env in "PVT_NO_CARRY=1" "PVT_X=1" "PVT_PIPE_WGS=3" "PVT_PIPE_WGS=4"; do
  echo "== $env"
  env $env timeout 600 python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3e  sustained %.3e  kernel_ms %.3f' % (d['value'], d['sustained']['value'], d['roofline']['kernel_ms_mean']), {k:'%.3e'%v['value'] for k,v in d.get('configs',{}).items()})
"
done
