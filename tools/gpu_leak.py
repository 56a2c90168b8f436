"""Developer tool (GPU box): 2000 create/close cycles of a device scene, then streams of launches: free device memory must not drift."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pvtrace_amd.engine import compile_scene, native
from pvtrace_amd.engine.emit import EmitterTables
from tests import scenes
scene = scenes.lsc_equivalent(); c = compile_scene(scene); em = EmitterTables(scene)
torch.cuda.init(); torch.zeros(1, device="cuda")
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
f0 = free()
for k in range(2000):
    d = native.DeviceScene(c, device=0, emitter=em); d.close()
f1 = free()
print("2000 DeviceScene create/close: delta MiB", (f0 - f1) / 2**20)
for k in range(500):
    d = native.DeviceScene(c, device=0, emitter=em); t = d.new_tallies(); d.trace(None, 1000, 1, t); torch.cuda.synchronize(); d.close()
torch.cuda.empty_cache(); f2 = free()
print("500 create/trace/close: delta MiB", (f1 - f2) / 2**20)
from pvtrace_amd import engine
for k in range(500):
    engine.simulate(scene, 1000, seed=k, record_every=0)
torch.cuda.empty_cache(); f3 = free()
print("500 simulate(): delta MiB", (f2 - f3) / 2**20)
for k in range(500):
    engine.simulate(scene, 1000, seed=k, record_every=1)
torch.cuda.empty_cache(); f4 = free()
print("500 simulate(record_every=1): delta MiB", (f3 - f4) / 2**20)
