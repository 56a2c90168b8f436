"""Developer tool (GPU box): where the end-to-end time of a packed-log simulate() goes (README case)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pvtrace_amd import engine
from tests import scenes
scene = scenes.hello_world()
engine.simulate(scene, 1000, seed=1, packed_log=True)
for rep in range(3):
    tic = time.perf_counter()
    r = engine.simulate(scene, 1_000_000, seed=1, packed_log=True)
    print(f"packed e2e {time.perf_counter() - tic:.4f} s  trace {r.elapsed*1e3:.2f} ms", flush=True)
    del r
pr = cProfile.Profile(); pr.enable()
r = engine.simulate(scene, 1_000_000, seed=1, packed_log=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
