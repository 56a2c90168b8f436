#!/bin/bash
# the round's measurement pass: tests, smoke, bench, kernel stats, PMC for cfg2/4/5, scene / mesh / e2e / history tables
set -x
bash tools/gpu_round3.sh
timeout 600 python tools/gpu_scene_times.py 2>&1 | tee gpurun_out/scene_times.txt | tail -12
timeout 900 python tools/gpu_mesh_times.py 2>&1 | tee gpurun_out/mesh_times.txt | tail -8
timeout 600 python tools/gpu_e2e.py 2>&1 | tee gpurun_out/e2e.txt | tail -14
timeout 600 python tools/gpu_history.py 2>&1 | tee gpurun_out/history_times.txt | tail -4
timeout 900 bash tools/gpu_history_pmc.sh 2>&1 | tee gpurun_out/history_pmc.txt | tail -30
timeout 300 bash tools/gpu_clock_overlap.sh 2>&1 | tee gpurun_out/clock_overlap.txt | tail -6
