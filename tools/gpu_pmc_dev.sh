#!/bin/bash
# Developer tool (GPU box): instruction-mix PMC pass of one or more dev builds on serial 4e6-photon launches.
# usage: tools/gpu_pmc_dev.sh name1 [name2 ...]   (build/dev/<name>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for name in "$@"; do
  PVT_LIB=$R/build/dev/$name.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
      --kernel-trace --output-format csv -d $R/gpurun_out/pmcdev_$name -o pmc -- python $R/tools/gpu_perf.py one > $R/gpurun_out/pmcdev_$name.log 2>&1
  python - "$name" <<PY
import csv, glob, sys, collections
name = sys.argv[1]
per = collections.defaultdict(list)
for path in glob.glob("$R/gpurun_out/pmcdev_%s/**/pmc_counter_collection.csv" % name, recursive=True):
    for r in csv.DictReader(open(path)):
        if "trace_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in per.items()}
n = 4e6
if c:
    print(f"{name}: VALU/photon {c['SQ_INSTS_VALU']/n:.1f} SALU {c['SQ_INSTS_SALU']/n:.1f} LDS {c['SQ_INSTS_LDS']/n:.1f} SMEM {c['SQ_INSTS_SMEM']/n:.2f} "
          f"lane-util {c['SQ_THREAD_CYCLES_VALU']/(64*c['SQ_ACTIVE_INST_VALU']):.3f} wait-inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f} wave-cycles {c['SQ_WAVE_CYCLES']/1e6:.0f}M")
else:
    print(name, "no counters")
PY
  grep lsc_ $R/gpurun_out/pmcdev_$name.log
done
