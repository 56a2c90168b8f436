#!/bin/bash
# Developer tool (GPU box): counters of the -DPVT_STATS=1 build for one 10^6-photon launch of a tile array (grid walk)
# usage: tools/gpu_grid_stats.sh lib.so [k ...]      (build/dev/<lib>.so made by tools/dev_build.sh <lib> -DPVT_STATS=1)
export PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$1; shift
for k in ${@:-11}; do
echo "== tiles$k  (PVT_GRID_CELLS=${PVT_GRID_CELLS:-default})"
K=$k python - <<'PY' 2>&1 | grep "pvt stats\|ms" | tail -5
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from benchmarks.configs import tiles_lsc
sc = tiles_lsc(int(os.environ["K"])); c = compile_scene(sc)
pos, d, wl, _ = emit_bundle(sc, 1_000_000, seed=5)
for rep in range(2):
    t = {}
    _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, 16, 0, 1, 0, timing=t)
print("kernel ms", t["kernel_ms"])
PY
done
