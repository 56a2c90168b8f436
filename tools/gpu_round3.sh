#!/bin/bash
# One GPU-box pass of round 3: tests, smoke, bench (default flags), rocprof kernel trace, PMC passes for
# cfg2 / cfg4 / cfg5.  usage: tools/gpu_round3.sh [skip-tests]
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "$1" != skip-tests ]; then
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest_gpu.log | tail -5
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-600
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --no-cpu-baseline --repeats 2 --sustained-s 0.2 --total-photons 0 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R
head -4 gpurun_out/prof/bench_kernel_stats.csv | cut -c1-200
for cfg in cfg2 cfg4 cfg5; do bash tools/gpu_pmc.sh pipelined $cfg > gpurun_out/pmc_$cfg.log 2>&1; done
