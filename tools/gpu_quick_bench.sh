#!/bin/bash
# quick: carry tests + default bench
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_two_ranks.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4e  first %.4e min %.4e max %.4e sustained %.4e strong %.4e kernel_ms %.3f' % (d['value'], d['repeats']['first_window'], d['repeats']['min'], d['repeats']['max'], d['sustained']['value'], d['strong_scaling']['value'], d['roofline']['kernel_ms_mean']))
print(d['strong_scaling']['predicted'])
for k,v in d['configs'].items(): print(k, '%.4e min %.4e max %.4e kernel_ms %.3f' % (v['value'], v['min'], v['max'], v['kernel_ms_mean']))
"
