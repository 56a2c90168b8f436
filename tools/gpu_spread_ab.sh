#!/bin/bash
# Round 6, VERDICT r5 "next" #4: the tail function's node tests side by side (PVT_TAIL_SPREAD, pvt_trace_kernel.h) against
# the same tree built without it (build/dev/spread0.so: -DPVT_TAIL_SPREAD=0) and with it from two nodes on
# (build/dev/spread2.so: -DPVT_TAIL_SPREAD_MIN_NODES=2), same box: parity first, then the lone wave's step, lone cfg4
# launches (the 1000-step photon), lone cfg2 launches, and the bench's stream.
# (The variant is kept as profiles/r06_coop_tail.patch, not in the tree: apply it first, then build spread0 / spread2 with tools/dev_build_full.sh.)
# usage: tools/gpu_spread_ab.sh [fuzz-count]
mkdir -p gpurun_out
{
echo "== parity, in-tree library (spread on, from 3 nodes)"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_carry.py tests/test_gpu_records.py -q -x 2>&1 | tail -3
echo "== parity, spread from 2 nodes"
PVT_LIB=$PWD/build/dev/spread2.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counters.py tests/test_gpu_carry.py tests/test_gpu_records.py -q -x 2>&1 | tail -3
echo "== fuzz (GPU vs oracle, bit for bit), in-tree library"
timeout 1500 python tools/gpu_fuzz.py 7000 ${1:-300} 2>&1 | tail -3
echo "== fuzz, spread from 2 nodes"
PVT_LIB=$PWD/build/dev/spread2.so timeout 1500 python tools/gpu_fuzz.py 9000 ${1:-300} 2>&1 | tail -3
for r in 1 2; do
for n in tree spread0 spread2; do
  lib=$PWD/build/dev/$n.so; [ $n = tree ] && lib=$PWD/pvtrace_amd/csrc/libpvtrace_hip.so
  echo "== lone step, $n (round $r)"; PVT_LIB=$lib timeout 300 python tools/gpu_lone_step.py 2>&1 | grep "per step"
done
done
for r in 1 2; do
for n in tree spread0 spread2; do
  lib=$PWD/build/dev/$n.so; [ $n = tree ] && lib=$PWD/pvtrace_amd/csrc/libpvtrace_hip.so
  echo "== lone cfg4 launches of 10^7, maxsteps 1000, $n (round $r)"; QUICK=1 REPS=9 PVT_LIB=$lib timeout 300 python tools/gpu_cfg4_tail.py 2>&1 | grep maxsteps
done
done
for n in tree spread0 spread2; do
  lib=$PWD/build/dev/$n.so; [ $n = tree ] && lib=$PWD/pvtrace_amd/csrc/libpvtrace_hip.so
  echo "== lone launches, $n"; PVT_LIB=$lib timeout 300 python tools/gpu_lone_launch.py 2>&1 | grep -v amdgpu.ids
done
one() {
  PVT_LIB=$2 python bench.py --extra-configs cfg4 --scene-sizes none 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs']['cfg4']
print('%-8s cfg2 value %.4e sustained %.4e strong %.4e | cfg4 window %.4e sustained %.4e' % ('$1', d['value'], d['sustained']['value'], d['strong_scaling']['value'], c['value'], c['sustained']['value']))"
}
for r in 1 2; do
  one tree $PWD/pvtrace_amd/csrc/libpvtrace_hip.so
  one spread0 $PWD/build/dev/spread0.so
done
} 2>&1 | tee gpurun_out/r06_spread_ab.txt
