#!/bin/bash
# Developer tool (GPU box): what bounds the mesh walk -- vector-memory (TA / TCP / TCC) and SQ counters of the trace
# kernel on one mesh scene of tools/gpu_mesh_stream.py (default ico5), each counter set in its own run with
# --kernel-trace only.  Output: gpurun_out/mesh_pmc_<scene>.txt (sums over the trace-kernel dispatches of 12 bundles)
scene=${1:-ico5}
# (PVT_LIB selects the build; PMC_TAG names the output, default the scene)
tag=${PMC_TAG:-$scene}
export TMPDIR=/tmp PVT_STREAM_STEPS=12
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
run() { name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/mesh_pmc_${tag}_$name -o pmc -- \
      python $R/tools/gpu_mesh_stream.py $scene > $R/gpurun_out/mesh_pmc_${tag}_$name.log 2>&1
}
# (the TA_* / TCP_* counter sets did not come back within ten minutes on this pool: run only on request)
if [ "$2" = with-ta ]; then
run ta TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
fi
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
run tcc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cd $R
python3 - "$tag" <<'PY' | tee gpurun_out/mesh_pmc_$tag.txt
import csv, glob, sys, collections
scene = sys.argv[1]
tot = collections.Counter(); n = 0
for f in glob.glob(f"gpurun_out/mesh_pmc_{scene}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n += 1
print(f"scene {scene}: trace-kernel dispatches {n} (10^6 photons each)")
for k, v in sorted(tot.items()): print(f"  {k:42s} {v:.4e}   per dispatch {v / max(n, 1):.4e}")
PY
