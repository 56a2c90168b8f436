#!/bin/bash
# long differential runs after the round's kernel changes
mkdir -p gpurun_out
timeout 1500 python tools/gpu_fuzz.py 500000 12000 2>&1 | tail -3 | tee gpurun_out/fuzz.txt
timeout 700 python tools/gpu_soak_carry.py 500 2>&1 | tail -3 | tee gpurun_out/soak_carry.txt
timeout 400 python tools/gpu_soak_api.py 240 2>&1 | tail -2 | tee gpurun_out/soak_api.txt
