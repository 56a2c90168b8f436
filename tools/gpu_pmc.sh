#!/bin/bash
# PMC passes for the bench's trace kernel (each counter set in its own run, with
# --kernel-trace only, as the pool rules require).  Outputs -> gpurun_out/pmc_*/
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o pmc -- \
      python $R/bench.py --gpus 1 --steps 5 --warmup 1 --streams 1 --no-cpu-baseline > $R/gpurun_out/pmc_$name.json 2> $R/gpurun_out/pmc_$name.err
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT
run grbm GRBM_GUI_ACTIVE
cd $R
ls gpurun_out/pmc_*; head -3 gpurun_out/pmc_fetch/*counter_collection.csv | cut -c1-400
