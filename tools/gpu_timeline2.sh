#!/bin/bash
# wave-level timelines of steady-state launches (dev build with -DPVT_TIMELINE=1), carry on/off, serial stream
export PVT_LIB=$GRAFT_REPO_ROOT/build/dev/timeline.so
F="--no-cpu-baseline --repeats 0 --sustained-s 0 --total-photons 0 --extra-configs none --spinup-s 0 --warmup 2 --steps 14 --streams 1"
rm -f /tmp/tl_*.bin
PVT_TIMELINE_FILE=/tmp/tl_carry.bin PVT_TIMELINE_FROM=8 timeout 300 python bench.py $F > /dev/null 2>&1
PVT_NO_CARRY=1 PVT_TIMELINE_FILE=/tmp/tl_nocarry.bin PVT_TIMELINE_FROM=8 timeout 300 python bench.py $F > /dev/null 2>&1
echo "=== carry (serial, 4 WG/CU)"; python tools/gpu_wave_timeline.py /tmp/tl_carry.bin | head -40
echo "=== no carry (serial, 4 WG/CU)"; python tools/gpu_wave_timeline.py /tmp/tl_nocarry.bin | head -24
