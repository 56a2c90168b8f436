"""Developer tool (GPU box): soak of the host-side machinery added in round 2 -- grouped bundles (tally sets), device
lists, worker threads -- against plain single-device calls, on random sizes.  usage: gpu_soak_api.py SECONDS"""
import os, sys, time, concurrent.futures
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from pvtrace_amd import engine
from tests import scenes
from tests.util import assert_bundles_identical

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(7)
makers = [scenes.bench_slab, scenes.lsc_equivalent, scenes.nested_cylinders, scenes.coated_slab, scenes.mesh_lsc]
tic = time.time(); rounds = 0
while time.time() - tic < budget:
    scene = makers[rounds % len(makers)]() if makers[rounds % len(makers)] is not scenes.bench_slab else scenes.bench_slab(recorders=True)
    total = int(rng.integers(1, 400_000)); bundle = int(rng.integers(1, 60_000)); seed = int(rng.integers(1, 2**31)); es = int(rng.integers(1, 2**31))
    whole = engine.simulate(scene, total, seed=seed, emit_seed=es, record_every=0, emission="device")
    def stream(**kw):
        acc = None; n = 0
        for r, traced in engine.simulate_stream(scene, total, bundle=bundle, seed=seed, emit_seed=es, record_every=0, emission="device", **kw):
            n += r.num_rays
            if acc is None: acc = {k: np.array(r.data[k], copy=True) for k in ("rec_distinct", "rec_crossings", "rec_bins", "rec_sums")}
            else:
                for k in acc: acc[k] += r.data[k]
        assert n == total == traced
        return acc
    a = stream(); b = stream(devices=[0, 0, 0])
    with concurrent.futures.ThreadPoolExecutor(2) as pool:
        c, d = [f.result() for f in (pool.submit(stream), pool.submit(stream, devices=[0, 0]))]
    two = engine.simulate(scene, total, seed=seed, emit_seed=es, record_every=0, emission="device", devices=[0, 0])
    for got in (a, b, c, d, two.data):
        for k in ("rec_distinct", "rec_crossings", "rec_bins"):
            assert np.array_equal(got[k], whole.data[k]), (rounds, k, total, bundle)
        assert np.allclose(got["rec_sums"], whole.data["rec_sums"], rtol=1e-10, atol=0), (rounds, total, bundle)
    rounds += 1
print(f"soak: {rounds} rounds in {time.time() - tic:.0f} s, all streams / device lists / threads equal to single calls", flush=True)
