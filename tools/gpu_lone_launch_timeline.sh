#!/bin/bash
# per-wave timeline of LONE 10^6-photon launches of the headline scene (dev build -DPVT_TIMELINE=1): when the cursor
# runs dry, when waves end, and on which SIMDs the last ones ran
# usage: tools/gpu_lone_launch_timeline.sh   (expects build/dev/timeline.so: tools/dev_build.sh timeline -DPVT_TIMELINE=1)
mkdir -p gpurun_out
export PVT_LIB=$PWD/build/dev/timeline.so PVT_TIMELINE_FILE=/tmp/tl.bin PVT_TIMELINE_FROM=3
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from benchmarks.configs import cfg2_lsc
scene = cfg2_lsc()
c = compile_scene(scene)
n = 1_000_000
pos, d, wl, _ = emit_bundle(scene, n, seed=1)
for rep in range(8):
    t = {}
    _kernel.trace_bundle(c, pos, d, wl, 11 + rep, 1000, 4, 0, 1, 0, timing=t)
    print("launch", rep, "kernel_ms", t["kernel_ms"], flush=True)
PY
python tools/gpu_wave_timeline.py /tmp/tl.bin v 2>&1 | tee gpurun_out/lone_launch_timeline.txt
