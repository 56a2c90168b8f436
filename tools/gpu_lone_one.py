"""Developer tool (GPU box): ONE launch shape for tools/gpu_lone_pmc.sh -- the totally reflected photon of
tools/gpu_lone_step.py in the headline slab, one lane, PVT_LONE_STEPS steps (default 4000), five launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pvtrace_amd as pv
from pvtrace_amd.engine import _kernel, compile_scene
from tests import scenes

steps = int(os.environ.get("PVT_LONE_STEPS", "4000"))
sc = scenes.lsc_equivalent()
slab = [n for n in sc.root.children if n.geometry is not None][0]
slab.geometry.material.components[1] = pv.Absorber(1e-9, name="Background")
c = compile_scene(sc)
v = np.array([0.6, 0.5, 0.62])
pos = np.array([[0.1, 0.2, 0.05]]); d = (v / np.linalg.norm(v))[None, :]; wl = np.array([800.0])
for rep in range(5):
    t = {}
    _kernel.trace_bundle(c, pos, d, wl, 1 + rep, steps, 4, 0, 1, 0, timing=t)
    print(f"{steps}-step launch {t['kernel_ms']:.3f} ms  = {t['kernel_ms'] / steps * 1e3:.3f} us per step", flush=True)
