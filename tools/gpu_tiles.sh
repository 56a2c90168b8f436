#!/bin/bash
# Scene-size scaling: the bench's own loop on the tiled scene family, one JSON line per size.
# usage (on the GPU box): tools/gpu_tiles.sh out_prefix [sizes...]
out=$1; shift
sizes=${@:-1 2 3 4 6 8 11}
mkdir -p gpurun_out
for k in $sizes; do
  python bench.py --config tiles$k --repeats 4 --steps 10 --sustained-s 1.5 --total-photons 0 --no-cpu-baseline \
      --extra-configs none --spinup-s 0.2 > gpurun_out/${out}_tiles$k.json 2> gpurun_out/${out}_tiles$k.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${out}_tiles$k.json"))
print("tiles$k", "median %.3e" % d["value"], "sustained %.3e" % d["sustained"]["value"], "kernel_ms %.3f" % d["roofline"]["kernel_ms_mean"], d["launch"])
PY
done
