"""Developer tool (GPU box): what a LONE cfg4 launch waits for.  Kernel time of single launches of nested_cylinders at
several sizes and step limits (`maxsteps`), with a `killed` recorder on both cylinders: if the tail of a 10^7-photon launch
is a handful of photons trapped by total internal reflection until the step limit ends them, its length follows the limit
and the recorder counts them.  (Seeds more than a launch apart: ray i of a launch draws from streams seed + i, so seeds one apart give the
same histories shifted by one ray.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from benchmarks.configs import cfg4_nested_cylinders
from pvtrace_amd.engine import Recorder, compile_scene, native
from pvtrace_amd.engine.compiler import EMIT_METHODS
from pvtrace_amd.engine.emit import EmitterTables

scene = cfg4_nested_cylinders()
for node in scene.root.levelorder():
    if node.name in ("A", "B"):
        node.recorders = list(node.recorders) + [Recorder(node.name + "-killed", event="killed")]
compiled = compile_scene(scene)
names = [s.name for s in compiled.recorder_specs]
dscene = native.DeviceScene(compiled, device=0, emitter=EmitterTables(scene, strict=True))
quick = os.environ.get("QUICK") == "1"
reps = int(os.environ.get("REPS", "6"))
for n in ((10_000_000,) if quick else (1_000_000, 10_000_000)):
    for maxsteps in ((1000,) if quick else (1000, 300, 100, 50)):
        times, killed = [], []
        for rep in range(reps):
            tallies = dscene.new_tallies()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            dscene.trace(None, n, 100 + 20_000_003 * rep, tallies, emit_seed=7 + 30_000_001 * rep, emit_method=EMIT_METHODS["kT"], maxsteps=maxsteps)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
            distinct = tallies["rec_distinct"].cpu().numpy()
            killed.append(sum(int(distinct[names.index(k)]) for k in ("A-killed", "B-killed")))
        print(f"n {n:.0e} maxsteps {maxsteps:5d}: launch ms {' '.join(f'{t:.3f}' for t in times[1:])}   killed {killed[1:]}   "
              f"mean {np.mean(times[1:]):.3f} median {np.median(times[1:]):.3f}", flush=True)
dscene.close()
