#!/bin/bash
# Developer tool: kernel timeline (start/end per dispatch) of a short sustained leg, with and without carried photons
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
flags="--gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --repeats 0 --sustained-s 0.05 --total-photons 0 --extra-configs none"
cd /tmp
for mode in carry nocarry; do
  [ $mode = nocarry ] && export PVT_NO_CARRY=1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$mode -o tl -- python $R/bench.py $flags > $R/gpurun_out/tl_$mode.json 2> $R/gpurun_out/tl_$mode.err
done
python $R/tools/timeline_summary.py $R/gpurun_out/tl_carry $R/gpurun_out/tl_nocarry
