"""The reference README's own engine figure (README.md:163-170): engine.simulate(scene, 1_000_000) on the
hello-world scene with DEFAULT arguments (record_every=1, max_events=128: a 15 GB event log)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pvtrace_amd import engine
from tests import scenes
scene = scenes.hello_world()
engine.simulate(scene, 1000, seed=1)  # warm
for n in (100_000, 1_000_000, 1_000_000, 1_000_000):   # (the first call of a size also allocates its pinned result blocks)
    tic = time.perf_counter()
    r = engine.simulate(scene, n, seed=1)
    wall = time.perf_counter() - tic
    print(f"n={n}: trace elapsed {r.elapsed*1e3:.1f} ms ({n/r.elapsed/1e6:.1f} M rays/s, reference convention: trace only), "
          f"kernel+memsets {r.kernel_ms:.1f} ms, end-to-end incl. emission and the download of the written rows (dense columns built on demand) {wall:.2f} s ({n/wall/1e6:.2f} M rays/s); "
          f"events {int(r.data['counts'].sum())}", flush=True)
    del r

for n in (1_000_000, 1_000_000):
    tic = time.perf_counter()
    r = engine.simulate(scene, n, seed=1, packed_log=True)
    wall = time.perf_counter() - tic
    print(f"n={n} packed_log=True: trace {r.elapsed*1e3:.1f} ms, end-to-end {wall:.3f} s ({n/wall/1e6:.2f} M rays/s); "
          f"rows {len(r.data['kind'])}", flush=True)
