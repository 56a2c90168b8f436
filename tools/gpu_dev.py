"""Developer loop (GPU box): parity of the headline scene against the oracle (history + tally modes)
followed by kernel timings.  PVT_LIB selects a dev build (tools/dev_build.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

names = sys.argv[1:] or ["lsc_equivalent"]
ok_all = True
for name in names:
    sc = scenes.ALL_SCENES[name]()
    c = compile_scene(sc)
    pos, d, wl, _ = emit_bundle(sc, 20000, seed=123)
    for rec_every, maxev, maxsteps, em in [(1, 64, 1000, 0), (0, 128, 1000, 0), (0, 128, 50, 1), (7, 16, 1000, 2)]:
        gpu = _kernel.trace_bundle(c, pos, d, wl, 42, maxsteps, maxev, em, 1, rec_every)
        cpu = O.trace_bundle(c, pos, d, wl, 42, maxsteps, maxev, em, 1, rec_every, math_mode=1)
        bad = [k for k in cpu if not (np.allclose(gpu[k], cpu[k], rtol=1e-12, atol=0) if k == "rec_sums"
                                      else np.array_equal(gpu[k], cpu[k]))]
        ok_all &= not bad
        print(f"{name:18s} rec_every={rec_every} maxsteps={maxsteps} emit={em}: {'OK' if not bad else 'FAIL ' + str(bad)}", flush=True)
        for k in bad[:3]:
            a, b = np.asarray(gpu[k]), np.asarray(cpu[k])
            idx = np.argwhere(a != b)
            print(f"    {k}: {len(idx)} differ; first {idx[:3].tolist()} gpu={a[tuple(idx[0])]} cpu={b[tuple(idx[0])]}")
print("PARITY", "OK" if ok_all else "FAILED", flush=True)
sc = scenes.lsc_equivalent(); c = compile_scene(sc)
for n in (1_000_000, 4_000_000):
    pos, d, wl, _ = emit_bundle(sc, n, seed=5)
    best = 1e9
    for rep in range(5):
        t = {}
        _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, 128, 0, 1, 0, timing=t)
        best = min(best, t["kernel_ms"])
    print(f"lsc n={n}: best {best:.3f} ms  {n / best / 1e3:.1f} M photons/s", flush=True)
