#!/bin/bash
# measured shader clock + wave-slot occupancy while three launches overlap (dev build -DPVT_TIMELINE=1)
export PVT_LIB=$GRAFT_REPO_ROOT/build/dev/timeline.so
F="--no-cpu-baseline --repeats 0 --sustained-s 0 --total-photons 0 --extra-configs none --warmup 3 --steps 60"
rm -f /tmp/tl.bin
PVT_TIMELINE_FILE=/tmp/tl.bin PVT_TIMELINE_FROM=700 timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench value (60-step window) %.4e' % d['value'])"
python tools/gpu_wave_timeline.py /tmp/tl.bin v | tail -16
