"""Developer tool (GPU box): wall time of engine.simulate() end to end (compile + emission + upload +
trace + download) against the trace-only figure, for host and device emission."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pvtrace_amd import engine
from tests import scenes

for name, make in (("lsc_equivalent", scenes.lsc_equivalent), ("nested_cylinders", scenes.nested_cylinders)):
    scene = make()
    for emission in ("host", "device"):
        for record_every in (0, 1000):
            engine.simulate(scene, 1000, seed=1, emission=emission, record_every=record_every)  # warm
            best = None
            for rep in range(3):
                tic = time.perf_counter()
                r = engine.simulate(scene, 1_000_000, seed=1 + rep, emission=emission, record_every=record_every)
                wall = time.perf_counter() - tic
                if best is None or wall < best[0]:
                    best = (wall, r.elapsed, r.kernel_ms)
            print(f"{name:18s} emission={emission:6s} record_every={record_every:<5d} wall {best[0]*1e3:8.1f} ms "
                  f"({1.0/best[0]:6.1f} M photons/s)  trace-only {best[1]*1e3:7.2f} ms  kernel {best[2]:6.2f} ms", flush=True)

scene = scenes.lsc_equivalent()
for bundle in (50_000, 200_000):
    for emission in ("device", "host"):
        list(engine.simulate_stream(scene, 100_000, bundle=bundle, seed=1, emission=emission, record_every=0))
        tic = time.perf_counter()
        total = 0
        nphot = 20_000_000 if emission == "device" else 2_000_000
        for result, traced in engine.simulate_stream(scene, nphot, bundle=bundle, seed=1, emission=emission,
                                                     record_every=0):
            total += int(result.data["rec_distinct"][7])
        wall = time.perf_counter() - tic
        print(f"simulate_stream {nphot:.0e} photons, bundle={bundle}, emission={emission}: {wall*1e3:.1f} ms "
              f"({nphot/1e6/wall:.1f} M photons/s)", flush=True)
