#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_carry.py -q -x --durations=10 -o faulthandler_timeout=45 > gpurun_out/r06_carry_debug.log 2>&1; tail -60 gpurun_out/r06_carry_debug.log
timeout 600 python tools/gpu_host_io.py 2>&1 | tee gpurun_out/r06_host_io.txt | tail -40
tools/gpu_spread_ab.sh 200 > /dev/null 2>&1; cat gpurun_out/r06_spread_ab.txt | tail -80
