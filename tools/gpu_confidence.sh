#!/bin/bash
# long differential runs on the final kernels of the round.  usage: tools/gpu_confidence.sh [fuzz scenes] [soak seconds]
mkdir -p gpurun_out
N=${1:-12000}; S=${2:-500}
timeout 2400 python tools/gpu_fuzz.py 700000 $N 2>&1 | tail -3 | tee gpurun_out/fuzz.txt
timeout $((S + 200)) python tools/gpu_soak_carry.py $S 2>&1 | tail -3 | tee gpurun_out/soak_carry.txt
timeout 400 python tools/gpu_soak_api.py 200 2>&1 | tail -2 | tee gpurun_out/soak_api.txt
