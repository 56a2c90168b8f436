#!/bin/bash
# Round 6, VERDICT r5 #9: does the 29.5 ms RECORD + host-rays launch reproduce?  (tools/gpu_e2e_outlier.py prints every repetition.)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python tools/gpu_e2e.py 2>&1 | tee gpurun_out/r06_e2e_first.txt | tail -14
timeout 900 python tools/gpu_e2e_outlier.py 5 2>&1 | tee gpurun_out/r06_e2e_outlier.txt | tail -60
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_outlier -o outlier -- python $R/tools/gpu_e2e_outlier.py 2 > $R/gpurun_out/r06_e2e_outlier_rocprof.txt 2>&1
cd $R
head -12 gpurun_out/prof_outlier/outlier_kernel_stats.csv | cut -c1-260
