"""Developer tool: the SAME rays and seeds through the tally kernel and the history kernel (record_every=1000), with and
without drain consolidation -- is the slow nested_cylinders launch a property of the photons or of a kernel variant?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from pvtrace_amd.engine import compile_scene, native
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
scene = scenes.nested_cylinders()
compiled = compile_scene(scene)
dev = torch.device("cuda", 0)
n = 1_000_000
for consolidate in (True, False):
    if not consolidate: os.environ["PVT_NO_CONSOLIDATE"] = "1"
    dscene = native.DeviceScene(compiled, device=0)
    print("drain consolidation", consolidate)
    for rep in (4, 5, 8, 11, 12, 13):
        pos, dirs, wl, _ = emit_bundle(scene, n, seed=200 + rep)
        rays = tuple(torch.from_numpy(a).to(dev) for a in (pos, dirs, wl))
        out = []
        for every in (0, 1000, 0, 1000):
            t = dscene.new_tallies()
            log = dscene.new_event_log(n, every, 128) if every else None
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dscene.trace(rays, n, 100 + rep, t, log=log, record_every=every, max_events=128)
            b.record(); torch.cuda.synchronize()
            out.append(a.elapsed_time(b))
        print(f"   rays {200+rep}: tally {out[0]:.3f} / {out[2]:.3f} ms   history(1000) {out[1]:.3f} / {out[3]:.3f} ms", flush=True)
    dscene.close()
