#!/bin/bash
# GPU-box pass for the history (event-log) path: parity tests that touch it, kernel times, HBM write traffic
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_records.py tests/test_gpu_parity.py tests/test_gpu_engine_api.py -x -q > gpurun_out/pytest_hist.log 2>&1; tail -5 gpurun_out/pytest_hist.log
timeout 600 python tools/gpu_history.py 2>&1 | tee gpurun_out/history_times.txt
timeout 600 python tools/gpu_history_scaling.py 2>&1 | tee gpurun_out/history_scaling.txt
timeout 900 bash tools/gpu_history_pmc.sh 2>&1 | tee gpurun_out/history_pmc.txt | tail -40
timeout 600 python tools/gpu_readme_case.py 2>&1 | tee gpurun_out/readme_case.txt
