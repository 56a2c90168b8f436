#!/bin/bash
# Developer tool (GPU box): the bench's loop (median window + sustained) on several configs, one line each
# usage: tools/gpu_configs_quick.sh tag cfg...
tag=$1; shift
for c in "$@"; do
  python bench.py --config $c --repeats 4 --steps 10 --sustained-s 1.5 --total-photons 0 --no-cpu-baseline --extra-configs none --spinup-s 0.2 > gpurun_out/${tag}_$c.json 2>gpurun_out/${tag}_$c.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_$c.json"))
print("$c", "median %.3e" % d["value"], "sustained %.3e" % d["sustained"]["value"], "kernel_ms %.3f" % d["roofline"]["kernel_ms_mean"], d["launch"])
PY
done
