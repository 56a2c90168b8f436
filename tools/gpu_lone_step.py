"""Developer tool (GPU box): latency of ONE wave's step -- the quantity a fenced window's closing drain is made of
(longest history x this).  A photon launched inside the headline slab along (0.6, 0.5, 0.62) at 800 nm is totally
reflected at every face for ever (background absorber set to 1e-9 cm^-1), so the kernel time of a one-wave launch is
maxsteps x the step latency; two values of maxsteps give the slope.  PVT_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pvtrace_amd.engine import _kernel, compile_scene
from tests import scenes

def slope(c, pos, d, wl, lo=2000, hi=6000, rec_every=0):
    ts = {}
    for ms in (lo, hi):
        best = 1e9
        for rep in range(3):
            t = {}
            _kernel.trace_bundle(c, pos, d, wl, 1 + rep, ms, 4, 0, 1, rec_every, timing=t)
            best = min(best, t["kernel_ms"])
        ts[ms] = best
    return (ts[hi] - ts[lo]) / (hi - lo) * 1e3, ts[lo]

import pvtrace_amd as pv
sc = scenes.lsc_equivalent()
slab = [n for n in sc.root.children if n.geometry is not None][0]
slab.geometry.material.components[1] = pv.Absorber(1e-9, name="Background")
c = compile_scene(sc)
for lanes in (1, 64):
    v = np.array([0.6, 0.5, 0.62])   # every direction cosine below cos(41.8 deg): total reflection at all six faces
    pos = np.tile(np.array([0.1, 0.2, 0.05]), (lanes, 1)); d = np.tile(v / np.linalg.norm(v), (lanes, 1)); wl = np.full(lanes, 800.0)
    if lanes > 1:   # spread the lanes a little so they are distinct histories of the same kind
        pos[:, 0] = np.linspace(-1, 1, lanes)
    us, base = slope(c, pos, d, wl)
    print(f"slab TIR, {lanes:2d} lane(s): {us:.3f} us per step  (2000-step launch {base:.3f} ms)", flush=True)
# a glass cylinder in an air sphere: a ray in the plane z = 0, 0.45 of the 0.5 radius off the axis, meets the barrel at
# 64 degrees for ever (whispering gallery) -- the history that bounds a fenced window of nested_cylinders
world = pv.Node(name="World", geometry=pv.Sphere(radius=10.0, material=pv.Material(refractive_index=1.0)))
pv.Node(name="A", parent=world, geometry=pv.Cylinder(length=2, radius=0.5, material=pv.Material(refractive_index=1.5)))
light = pv.Node(name="Light", parent=world, light=pv.Light())
c = compile_scene(pv.Scene(world))
for lanes in (1, 64):
    pos = np.tile(np.array([0.45, 0.0, 0.0]), (lanes, 1)); d = np.tile(np.array([0.0, 1.0, 0.0]), (lanes, 1))
    wl = np.full(lanes, 555.0)
    if lanes > 1:
        pos[:, 2] = np.linspace(-0.9, 0.9, lanes)
    us, base = slope(c, pos, d, wl)
    print(f"cylinder, whispering ray, {lanes:2d} lane(s): {us:.3f} us per step  (2000-step launch {base:.3f} ms)", flush=True)
# BASELINE configs[3] itself: a photon of nested_cylinders that total internal reflection holds until `maxsteps` ends it
# (found among 2 10^5 traced with histories: the first whose log fills) -- the history a fenced cfg4 window waits for
from benchmarks.configs import cfg4_nested_cylinders
from pvtrace_amd.engine.emit import emit_bundle
sc = cfg4_nested_cylinders()
c = compile_scene(sc)
pos, d, wl, _ = emit_bundle(sc, 200_000, seed=3)
out = _kernel.trace_bundle(c, pos, d, wl, 5, 600, 2, 0, 1, 0)
names = list(c.recorder_names)
t = {}
res = _kernel.trace_bundle(c, pos, d, wl, 5, 600, 700, 0, 1, 1)
long_ones = np.flatnonzero(res["counts"] >= 600)
print(f"cfg4: {len(long_ones)} of 200000 photons are still alive after 600 steps", flush=True)
if len(long_ones):
    j = int(long_ones[0])
    rows = slice(j * 700, j * 700 + 8)
    print("   its first events:", res["kind"][rows].tolist(), "hit", res["hit"][rows].tolist(), "container", res["container"][rows].tolist())
    for lanes in (1, 64):
        pick = long_ones[:lanes] if len(long_ones) >= lanes else np.resize(long_ones, lanes)
        # every lane the SAME photon (same ray, same seed offset is not possible: the stream is seed + index) -- so lane k
        # runs ray pick[k] with the stream that trapped it: seed 5 + its index; a bundle of one ray at ray_offset = index
        if lanes == 1:
            ts = {}
            for ms in (2000, 6000):
                best = 1e9
                for rep in range(3):
                    tt = {}
                    _kernel.trace_bundle(c, pos[j:j + 1], d[j:j + 1], wl[j:j + 1], 5, ms, 4, 0, 1, 0, ray_offset=j, timing=tt)
                    best = min(best, tt["kernel_ms"])
                ts[ms] = best
            print(f"cfg4 trapped photon,  1 lane(s): {(ts[6000] - ts[2000]) / 4000 * 1e3:.3f} us per step  (2000-step launch {ts[2000]:.3f} ms)", flush=True)
