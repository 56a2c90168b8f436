"""Developer tool (GPU box): the launch VERDICT r5 #9 asked about -- `engine.simulate(lsc_equivalent, 10^6, emission="host",
record_every=1000)` read 29.5 ms of "kernel" time in profiles/r05_e2e.txt where its twins read 0.7-0.8 ms.

Every repetition is printed (not the best of three), with the numpy seed fixed and printed, and the same launch is timed
three ways so that host delay between the two HIP events can be told from time on the GPU:
  A. engine.simulate            -- `kernel_ms` = HIP events around DeviceScene.trace (what r05_e2e.txt printed)
  B. DeviceScene.trace directly -- rays resident, event log allocated and the device idle before the first event
  C. the same launch under the kernel's own always-on counters (trips of the photon loop, photon steps)
usage: python tools/gpu_e2e_outlier.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from pvtrace_amd import engine
from pvtrace_amd.engine import api, native
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = 1_000_000
scene = scenes.lsc_equivalent()
print(f"numpy seed fixed per repetition: np.random.seed(100 + rep); kernel seed 1 + rep; N = {N}", flush=True)

for emission, record_every in (("host", 0), ("host", 1000), ("device", 1000), ("host", 1000)):
    np.random.seed(99)
    engine.simulate(scene, 1000, seed=1, emission=emission, record_every=record_every)   # warm
    for rep in range(reps):
        np.random.seed(100 + rep)
        tic = time.perf_counter()
        r = engine.simulate(scene, N, seed=1 + rep, emission=emission, record_every=record_every)
        wall = time.perf_counter() - tic
        print(f"A simulate   emission={emission:6s} record_every={record_every:<5d} rep {rep}: wall {wall*1e3:8.2f} ms  "
              f"elapsed {r.elapsed*1e3:8.3f} ms  kernel_ms {r.kernel_ms:8.3f}  events logged {int(r.data['counts'].sum())}",
              flush=True)

# B: the launch on its own
compiled = engine.compile_scene(scene)
dscene = native.DeviceScene(compiled, device=0)
dev = torch.device("cuda", 0)
for record_every in (0, 1000, 1000):
    for rep in range(reps):
        np.random.seed(100 + rep)
        pos, direc, wl, _ = emit_bundle(scene, N, seed=None)
        rays = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pos, direc, wl))
        tallies = dscene.new_tallies()
        log = dscene.new_event_log(N, record_every, 128) if record_every else None
        torch.cuda.synchronize()
        dscene.counters(reset=True)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        start.record()
        dscene.trace(rays, N, 1 + rep, tallies, log=log, record_every=record_every, max_events=128, log_prefill=False)
        t1 = time.perf_counter()
        stop.record()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        c = dscene.counters()
        print(f"B trace      record_every={record_every:<5d} rep {rep}: events {start.elapsed_time(stop):8.3f} ms  "
              f"host in trace() {1e3*(t1-t0):7.3f} ms  launch->idle {1e3*(t2-t0):7.3f} ms  counters {c}", flush=True)
dscene.close()
