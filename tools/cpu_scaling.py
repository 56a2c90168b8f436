"""Host-core scaling probe of the CPU referee on the GPU box (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from pvtrace_amd.engine import compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ; cat /proc/loadavg")
sc = scenes.lsc_equivalent(); c = compile_scene(sc)
pos, d, wl, _ = emit_bundle(sc, 2_000_000, seed=1)
for thr in (1, 8, 16, 32, 64, 128, 256):
    if thr > os.cpu_count(): break
    n = min(2_000_000, 200_000 * thr)
    O.trace_bundle(c, pos[:20000], d[:20000], wl[:20000], 1, 1000, 128, 0, thr, 0)
    tic = time.perf_counter(); O.trace_bundle(c, pos[:n], d[:n], wl[:n], 1, 1000, 128, 0, thr, 0); el = time.perf_counter() - tic
    print(f"threads={thr:4d} n={n} {n/el/1e6:.2f} M photons/s")
