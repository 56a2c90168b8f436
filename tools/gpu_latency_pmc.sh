#!/bin/bash
# Developer tool (GPU box): average LDS / scalar / vector memory latency seen by the trace kernel (derived PMC metrics),
# one metric per pass, serial 4e6-photon launches (tools/gpu_perf.py one)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for m in LdsLatency SmemLatency VmemLatency; do
  timeout 300 rocprofv3 --pmc $m --kernel-trace --output-format csv -d $R/gpurun_out/lat_$m -o pmc -- python $R/tools/gpu_perf.py one > $R/gpurun_out/lat_$m.log 2>&1
  python - "$m" <<PY
import csv, glob, sys
m = sys.argv[1]
vals = []
for path in glob.glob("$R/gpurun_out/lat_%s/**/pmc_counter_collection.csv" % m, recursive=True):
    for r in csv.DictReader(open(path)):
        if "trace_kernel" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
print(m, [round(v, 1) for v in vals[-3:]])
PY
done
