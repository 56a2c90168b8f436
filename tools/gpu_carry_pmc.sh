#!/bin/bash
# Developer tool: instruction counters per launch of the bench with and without carried photons
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
flags="--gpus 1 --steps 9 --warmup 1 --no-cpu-baseline --repeats 0 --sustained-s 0 --total-photons 0 --spinup-s 0 --ray-buffers 2 --extra-configs none"
cd /tmp
for mode in carry nocarry; do
  [ $mode = nocarry ] && export PVT_NO_CARRY=1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/cpmc_$mode -o pmc -- python $R/bench.py $flags > $R/gpurun_out/cpmc_$mode.json 2> $R/gpurun_out/cpmc_$mode.err
done
python - <<PY
import csv, glob, collections
for mode in ("carry", "nocarry"):
    per = collections.defaultdict(dict)
    for path in glob.glob("$R/gpurun_out/cpmc_%s/**/pmc_counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(path)):
            if "trace_kernel" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                per[int(r["Dispatch_Id"])]["grid"] = int(r["Grid_Size"]) // 256
    print(mode)
    for d in sorted(per):
        c = per[d]
        print(f"  dispatch {d:4d} grid {c['grid']:5d} waves {c['SQ_WAVES']:7.0f} VALU {c['SQ_INSTS_VALU']/1e6:8.2f}M  SALU {c['SQ_INSTS_SALU']/1e6:7.2f}M LDS {c['SQ_INSTS_LDS']/1e6:6.2f}M lane-util {c['SQ_THREAD_CYCLES_VALU']/(64*c['SQ_ACTIVE_INST_VALU']):.3f} wave-cycles {c['SQ_WAVE_CYCLES']/1e6:8.1f}M")
PY
