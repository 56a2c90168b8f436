#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
timeout 900 python tools/gpu_fuzz.py 900000 6000 2>&1 | tail -2 | tee gpurun_out/fuzz2.txt
bash tools/gpu_quick_bench.sh 2>&1 | tail -6
python tools/gpu_scene_times.py nested_cylinders hello_world kitchen_sink fresnel_box 2>&1 | tail -4
