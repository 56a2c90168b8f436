#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
timeout 600 python tools/gpu_anomaly.py 2>&1 | tee gpurun_out/anomaly.txt | tail -50
timeout 600 python tools/gpu_e2e.py 2>&1 | tee gpurun_out/e2e.txt | tail -20
