#!/bin/bash
# PMC passes for the bench's trace kernel (each counter set in its own run, with --kernel-trace only, as the
# pool rules require).  usage: tools/gpu_pmc.sh [pipelined|serial] [cfg2|cfg4|cfg5]   Outputs -> gpurun_out/pmc_<mode>[_cfgN]_*/
# "pipelined" is the bench's default configuration (3 bundles in flight, 2 workgroups per CU per launch);
# "serial" is --streams 1 (4 workgroups per CU).  NB rocprofv3 serialises dispatches while it samples
# counters, so the pipelined passes measure the pipelined LAUNCH SHAPE, one launch at a time.
set -x
mode=${1:-pipelined}
cfg=${2:-cfg2}
flags="--gpus 1 --steps 6 --warmup 1 --no-cpu-baseline --repeats 0 --sustained-s 0 --total-photons 0 --spinup-s 0 --ray-buffers 2 --extra-configs none --config $cfg"
[ "$mode" = serial ] && flags="$flags --streams 1"
[ "$cfg" != cfg2 ] && mode=${mode}_$cfg
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${mode}_$name -o pmc -- \
      python $R/bench.py $flags > $R/gpurun_out/pmc_${mode}_$name.json 2> $R/gpurun_out/pmc_${mode}_$name.err
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT
run grbm GRBM_GUI_ACTIVE
cd $R
ls -d gpurun_out/pmc_${mode}_*/ ; head -3 gpurun_out/pmc_${mode}_fetch/*counter_collection.csv | cut -c1-300
