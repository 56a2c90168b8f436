#!/bin/bash
# Developer tool (GPU box): the cycle budget of a lone wave's step from the SQ counters -- tools/gpu_lone_one.py (one
# totally reflected photon, 4000 steps per launch) under rocprofv3 --pmc, each counter set in its own run with
# --kernel-trace only.  Output: gpurun_out/lone_pmc.txt (per STEP: wave-instructions by kind, quad-cycle counters x 4)
export TMPDIR=/tmp PVT_LONE_STEPS=${PVT_LONE_STEPS:-4000}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/lone_pmc_$name -o pmc -- \
      python $R/tools/gpu_lone_one.py > $R/gpurun_out/lone_pmc_$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT
run sq4 SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INST_LEVEL_LDS
cd $R
python3 - <<'PY' | tee gpurun_out/lone_pmc.txt
import csv, glob, collections, os
steps = int(os.environ["PVT_LONE_STEPS"])
tot = collections.Counter(); n = collections.Counter()
for f in glob.glob("gpurun_out/lone_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(f"per step of a lone wave (sums over the launch / {steps} steps; every launch also starts and retires its other, idle waves)")
for k in sorted(tot):
    print(f"{k:28s} {tot[k] / n[k] / steps:12.2f}   ({n[k]} dispatches)")
PY
cat gpurun_out/lone_pmc_sq1.log | tail -5
