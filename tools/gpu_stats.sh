#!/bin/bash
# Developer tool (GPU box): the -DPVT_STATS=1 build's counters (wave-iterations, live lanes, lanes per class, section clocks of a lone wave) for one 10^6-photon launch of cfg2
export PVT_LIB=$GRAFT_REPO_ROOT/build/dev/stats.so
python - <<'PY' 2>&1 | grep "pvt stats" | tail -6
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from benchmarks.configs import cfg2_lsc
sc = cfg2_lsc(); c = compile_scene(sc)
pos, d, wl, _ = emit_bundle(sc, 1_000_000, seed=5)
for rep in range(2):
    _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, 16, 0, 1, 0)
PY
