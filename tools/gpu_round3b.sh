#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-300
timeout 600 python tools/gpu_history.py 2>&1 | tee gpurun_out/history_times.txt
timeout 600 python tools/gpu_readme_case.py 2>&1 | tee gpurun_out/readme_case.txt
