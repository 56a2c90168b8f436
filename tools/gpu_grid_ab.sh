#!/bin/bash
# Developer tool (GPU box): kernel time of single 2*10^6-photon launches on tile arrays for several dev builds
# usage: tools/gpu_grid_ab.sh "lib1.so lib2.so" "k1 k2"
for lib in $1; do
for k in $2; do
PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$lib K=$k python - <<'PY' 2>&1 | grep -v "pvt stats" | tail -1
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from benchmarks.configs import tiles_lsc
k = int(os.environ["K"])
sc = tiles_lsc(k); c = compile_scene(sc)
n = 2_000_000
pos, d, wl, _ = emit_bundle(sc, n, seed=5)
best = 1e9
for rep in range(5):
    t = {}
    _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, 16, 0, 1, 0, timing=t)
    best = min(best, t["kernel_ms"])
print(f"{os.path.basename(os.environ['PVT_LIB']):24s} cells={os.environ.get('PVT_GRID_CELLS','default'):8s} tiles{k}: {best:.3f} ms  {n / best / 1e3:.0f} M photons/s")
PY
done; done
