#!/bin/bash
# Developer tool (GPU box): HBM traffic of history-mode launches (tools/gpu_history_scaling.py) -- FETCH_SIZE / WRITE_SIZE passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/hist_pmc_$c -o pmc -- python $R/tools/gpu_history_scaling.py > $R/gpurun_out/hist_pmc_$c.log 2>&1
done
python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for path in glob.glob("$R/gpurun_out/hist_pmc_%s/**/pmc_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(path)):
            if "trace_kernel" in r["Kernel_Name"] or "unpack_log" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:60], float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    rows.sort()
    print(c)
    for d, k, v, ms in rows:
        print(f"  dispatch {d:3d} {k:60s} {v/1024/1024:8.2f} GiB (KiB units)   {ms:7.3f} ms")
PY
