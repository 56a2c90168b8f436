#!/bin/bash
# Developer tool (GPU box): bench headline/sustained over pipeline depth x workgroups per CU
for streams in 2 3 4 5; do for wgs in 1 2 3 4; do
  PVT_PIPE_WGS=$wgs timeout 300 python bench.py --streams $streams --no-cpu-baseline --total-photons 0 --repeats 6 --sustained-s 0.5 --extra-configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('streams $streams wgs/cu $wgs', 'headline %.3fe9 median %.3fe9 sustained %.3fe9' % (d['value']/1e9, d['repeats']['median']/1e9, d['sustained']['value']/1e9))"
done; done
