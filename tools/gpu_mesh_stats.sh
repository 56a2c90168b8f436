#!/bin/bash
# Developer tool (GPU box): the -DPVT_STATS=1 mesh build's walk counters (box trips, triangle trips, lanes in each, boxes and
# triangles per lane-step) for single 10^6-photon launches of the scenes of tools/gpu_mesh_stream.py.
# build first: DEVV=2 tools/dev_build.sh mstats -DPVT_STATS=1
export PVT_LIB=$GRAFT_REPO_ROOT/build/dev/mstats.so
for s in ${@:-mesh_gem ico3 ico5 ico7}; do
echo "== $s"
python - $s <<'PY' 2>&1 | grep "pvt stats" | tail -4
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import pvtrace_amd as pv
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
name = sys.argv[1]
def ico(sub):
    sc = scenes.hello_world()
    ball = [n for n in sc.root.children if n.geometry is not None][0]
    ball.geometry = pv.Mesh.icosphere(sub, 1.0, material=ball.geometry.material)
    return sc
sc = ico(int(name[3:])) if name.startswith("ico") else getattr(scenes, name)()
c = compile_scene(sc)
pos, d, wl, _ = emit_bundle(sc, 1_000_000, seed=5)
_kernel.trace_bundle(c, pos, d, wl, 1, 1000, 16, 0, 1, 0)
PY
done
