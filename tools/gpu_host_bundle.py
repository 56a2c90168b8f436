"""pvt_trace_bundle with HOST arrays (the literal stand-in for _kernel.trace_bundle): where its milliseconds go, and what
the chunked upload buys.  PVT_HOST_PHASES=1 makes the library print its phases; PVT_HOST_CHUNK_RAYS sets the chunk."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from benchmarks.configs import cfg2_lsc
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle

scene = cfg2_lsc()
compiled = compile_scene(scene)
for n in (1_000_000, 4_000_000):
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=1)
    _kernel.trace_bundle(compiled, pos[:1000], dirs[:1000], wl[:1000], 1, 1000, 128, 0, 1, 0)
    for chunk in ("0", None, "131072", "262144"):
        if chunk is None:
            os.environ.pop("PVT_HOST_CHUNK_RAYS", None)
        else:
            os.environ["PVT_HOST_CHUNK_RAYS"] = chunk
        times = []
        for rep in range(9):
            if rep == 8:
                os.environ["PVT_HOST_PHASES"] = "1"
            timing = {}
            t = time.perf_counter()
            out = _kernel.trace_bundle(compiled, pos, dirs, wl, 1 + rep, 1000, 128, 0, 1, 0, timing=timing)
            times.append((time.perf_counter() - t, timing["kernel_ms"]))
            os.environ.pop("PVT_HOST_PHASES", None)
        b = min(times)
        label = "one upload, one launch" if chunk == "0" else f"chunks of {chunk or 'default (524288)'}"
        print(f"pvt_trace_bundle host arrays n={n} {label}: best {b[0]*1e3:.2f} ms ({n/b[0]/1e6:.0f} M photons/s), "
              f"upload+trace span {b[1]:.2f} ms; all: " + " ".join(f"{t*1e3:.2f}" for t, _ in times), flush=True)
os.environ.pop("PVT_HOST_CHUNK_RAYS", None)
# with an event log sampled like the reference's harness (record_every = 1000)
n = 2_000_000
pos, dirs, wl, _ = emit_bundle(scene, n, seed=2)
for chunk in ("0", None):
    if chunk is None:
        os.environ.pop("PVT_HOST_CHUNK_RAYS", None)
    else:
        os.environ["PVT_HOST_CHUNK_RAYS"] = chunk
    times = []
    for rep in range(5):
        t = time.perf_counter()
        _kernel.trace_bundle(compiled, pos, dirs, wl, 1 + rep, 1000, 128, 0, 1, 1000)
        times.append(time.perf_counter() - t)
    print(f"pvt_trace_bundle host arrays n={n} record_every=1000 max_events=128 {'one upload' if chunk else 'default chunks'}: "
          f"best {min(times)*1e3:.2f} ms ({n/min(times)/1e6:.0f} M photons/s)", flush=True)
