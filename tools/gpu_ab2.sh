#!/bin/bash
# sweep: carry on/off x streams x workgroups per CU (sustained leg 1 s)
mkdir -p gpurun_out
F="--no-cpu-baseline --repeats 2 --sustained-s 1.0 --total-photons 0 --extra-configs none"
run() { # label, env..., -- flags
  label=$1; shift
  env "$@" timeout 600 python bench.py $F $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: value %.3e  sustained %.3e  kernel_ms %.3f' % (d['value'], d['sustained']['value'], d['roofline']['kernel_ms_mean']))
"
}
for streams in 1 2 3 4; do
  for wgs in 1 2 4; do
    EXTRA="--streams $streams"
    run "carry   streams $streams wgs $wgs" PVT_PIPE_WGS=$wgs
    run "nocarry streams $streams wgs $wgs" PVT_PIPE_WGS=$wgs PVT_NO_CARRY=1
  done
done
