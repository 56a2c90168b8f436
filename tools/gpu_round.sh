#!/bin/bash
# One GPU-box pass: tests, smoke, bench (default flags), serial bench, rocprof kernel trace of both
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest_gpu.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-400
timeout 600 python bench.py --streams 1 --no-cpu-baseline 2>>gpurun_out/bench.err | tee gpurun_out/bench_serial.json | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --no-cpu-baseline --repeats 2 --sustained-s 0.2 --total-photons 0 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -o bench -- python $R/bench.py --streams 1 --no-cpu-baseline --repeats 2 --sustained-s 0.2 --total-photons 0 > $R/gpurun_out/prof_bench_serial.json 2> $R/gpurun_out/prof_serial.err
cd $R
head -3 gpurun_out/prof/bench_kernel_stats.csv | cut -c1-200; head -3 gpurun_out/prof_serial/bench_kernel_stats.csv | cut -c1-200
