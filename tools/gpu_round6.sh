#!/bin/bash
# One GPU-box pass of round 6 (successor of tools/gpu_round5.sh): tests, smoke, the bench with the driver's flags, rocprofv3
# --kernel-trace --stats of the same command, the per-scene / mesh / end-to-end / history / lone-step tables.
# PMC passes: tools/gpu_round6_pmc.sh.   usage: tools/gpu_round6.sh [skip-tests]
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "$1" != skip-tests ]; then
t0=$(date +%s)
PVT_EIGHT_RANKS_RECORD=$R/gpurun_out/eight_ranks.json timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest -m gpu: rc $? wall $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rror" gpurun_out/pytest_gpu.log | tail -5
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --no-cpu-baseline --repeats 2 --sustained-s 0.2 --total-photons 0 --extra-configs none > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R
head -4 gpurun_out/prof/bench_kernel_stats.csv | cut -c1-200
timeout 600 python tools/gpu_scene_times.py 2>&1 | tee gpurun_out/scene_times.txt | tail -12
timeout 900 python tools/gpu_mesh_times.py 2>&1 | tee gpurun_out/mesh_times.txt | tail -8
PVT_STREAM_STEPS=150 timeout 900 python tools/gpu_mesh_stream.py 2>&1 | tee gpurun_out/mesh_stream.txt | tail -8
timeout 600 python tools/gpu_e2e.py 2>&1 | tee gpurun_out/e2e.txt | tail -14
timeout 600 python tools/gpu_history.py 2>&1 | tee gpurun_out/history_times.txt | tail -4
timeout 300 python tools/gpu_lone_step.py 2>&1 | tee gpurun_out/lone_step.txt | tail -4
timeout 300 python tools/gpu_lone_launch.py 2>&1 | tee gpurun_out/lone_launch.txt | tail -4
