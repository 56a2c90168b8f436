#!/bin/bash
# Developer tool (GPU box): SQ counters of single launches of the headline scene at several sizes (tools/gpu_perf.py lsc): the bulk of a launch against its tail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/bulk_pmc -o pmc -- python $R/tools/gpu_perf.py lsc > $R/gpurun_out/bulk_pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/bulk_pmc2 -o pmc -- python $R/tools/gpu_perf.py lsc > $R/gpurun_out/bulk_pmc2.log 2>&1
cd $R; grep -v amdgpu gpurun_out/bulk_pmc.log | grep lsc_
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed" 
