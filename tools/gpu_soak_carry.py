"""Developer tool (GPU box): soak of the photons carried between launches (PVT_FLAG_CARRY_OUT, BundlePipeline) on random
scenes (tests/fuzz.py, extensions included), random job sizes, bundle sizes and pipeline depths, against the CPU referee's
totals for the whole job.  usage: gpu_soak_carry.py SECONDS"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle as O
from pvtrace_amd.engine import BundlePipeline, compile_scene, native
from pvtrace_amd.engine.emit import emit_bundle
from tests.fuzz import random_scene

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(11)
dev = torch.device("cuda", 0)
tic = time.time(); rounds = 0; launches = 0; photons = 0
while time.time() - tic < budget:
    scene = random_scene(1000 + rounds, extensions=bool(rounds % 2))
    try:
        c = compile_scene(scene)
    except Exception:
        rounds += 1; continue
    total = int(rng.integers(1, 150_000)); seed = int(rng.integers(1, 2**31)); maxsteps = int(rng.choice([40, 300, 1000]))
    method = int(rng.integers(0, 3))
    pos, dirs, wl, _ = emit_bundle(scene, total, seed=rounds)
    cpu = O.trace_bundle(c, pos, dirs, wl, seed, maxsteps, 8, method, 8, 0, math_mode=O.MATH_PORTABLE)
    rays = tuple(torch.from_numpy(a).to(dev) for a in (pos, dirs, wl))
    dscene = native.DeviceScene(c, device=0)
    try:
        for depth in (1, int(rng.integers(2, 4))):
            pipe = BundlePipeline(dscene, depth=depth, carry=True)
            at = 0
            edges = [0]
            while edges[-1] < total:
                edges.append(min(total, edges[-1] + int(rng.integers(1, max(2, total // 2)))))
            for k, (a, b) in enumerate(zip(edges[:-1], edges[1:])):
                last = b >= total
                pipe.submit(tuple(t[a:b] for t in rays), b - a, seed=seed, ray_offset=a, maxsteps=maxsteps, emit_method=method,
                            timed=False, tail=last and bool(rng.integers(0, 2)), closing=bool(rng.integers(0, 4) == 0))
                launches += 1
                if rng.integers(0, 6) == 0 and not last:      # a reader in the middle of the job: totals so far, then on
                    pipe.reduce_totals()
            got = pipe.totals_host()
            for key in ("rec_distinct", "rec_crossings", "rec_bins"):
                assert np.array_equal(got[key], cpu[key]), (rounds, depth, key, total, edges)
            assert np.allclose(got["rec_sums"], cpu["rec_sums"], rtol=1e-10, atol=0, equal_nan=True), (rounds, depth)
            photons += total
    finally:
        dscene.close()
    rounds += 1
print(f"carry soak: {rounds} random scenes, {launches} launches, {photons} photons in {time.time() - tic:.0f} s: every job's totals equal the referee's", flush=True)
