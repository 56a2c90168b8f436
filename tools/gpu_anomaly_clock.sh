#!/bin/bash
# per-wave timelines of the launches of tools/gpu_anomaly.py's host-emission case (dev build -DPVT_TIMELINE=1)
export PVT_LIB=$GRAFT_REPO_ROOT/build/dev/timeline.so
rm -f /tmp/tla.bin
PVT_TIMELINE_FILE=/tmp/tla.bin python - <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pvtrace_amd import engine
from tests import scenes
scene = scenes.nested_cylinders()
for rep in range(14):
    r = engine.simulate(scene, 1_000_000, seed=100 + rep, emit_seed=200 + rep, emission="host", record_every=1000)
    print(f"host emission launch {rep}: kernel {r.kernel_ms:.3f} ms  longest sampled history {int(r.data['counts'].max())}", flush=True)
PY
python tools/gpu_wave_timeline.py /tmp/tla.bin v | grep "^launch" | cut -c1-220
