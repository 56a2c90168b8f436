#!/bin/bash
# Developer tool (GPU box): instruction-cache counters of single launches of the headline scene (tools/gpu_perf.py lsc)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/icache_pmc -o pmc -- python $R/tools/gpu_perf.py lsc > $R/gpurun_out/icache_pmc.log 2>&1
tail -3 $R/gpurun_out/icache_pmc.log
