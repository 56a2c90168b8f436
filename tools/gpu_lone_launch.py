"""Developer tool (GPU box): kernel time of LONE launches -- one job of BASELINE's literal size, nothing in flight beside it
(VERDICT r4 #1): cfg2 at 10^6 photons (array input) and cfg4 / cfg5 at 10^7 (device emission), best and median of a few
launches each, through the device-resident entry.  PVT_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from benchmarks.configs import CONFIGS
from pvtrace_amd.engine import compile_scene, native
from pvtrace_amd.engine.compiler import EMIT_METHODS
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle

dev = torch.device("cuda", 0)
for name, n in (("cfg2", 1_000_000), ("cfg4", 10_000_000), ("cfg5", 10_000_000)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    spec = CONFIGS[name]
    scene = spec["build"]()
    compiled = compile_scene(scene)
    rays = None
    try:
        if name == "cfg2":
            dscene = native.DeviceScene(compiled, device=0)
            rays = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in emit_bundle(scene, n, seed=1)[:3])
        else:
            dscene = native.DeviceScene(compiled, device=0, emitter=EmitterTables(scene, strict=True))
    except Exception as exc:   # (a developer build without the device-emission variants)
        print(f"{name}: skipped ({exc})")
        continue
    times = []
    for rep in range(12):
        tallies = dscene.new_tallies()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        try:
            dscene.trace(rays, n, 100 + rep, tallies, emit_seed=7 + rep, emit_method=EMIT_METHODS[spec["emit_method"]])
        except Exception as exc:
            print(f"{name}: skipped ({exc})")
            times = None
            break
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    if times:
        t = sorted(times[2:])
        print(f"{name}: lone {n:.0e}-photon launch  best {t[0]:.3f} ms  median {t[len(t) // 2]:.3f} ms  worst {t[-1]:.3f} ms  "
              f"-> {n / t[len(t) // 2] / 1e3:.0f} M photons/s", flush=True)
    dscene.close()
