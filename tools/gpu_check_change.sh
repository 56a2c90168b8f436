#!/bin/bash
# after a kernel change: the GPU tests, a fuzz run, scene times and the quick bench
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
timeout 900 python tools/gpu_fuzz.py 1300000 ${1:-3000} 2>&1 | tail -2
python tools/gpu_scene_times.py 2>&1 | tail -7
bash tools/gpu_quick_bench.sh 2>&1 | tail -5
