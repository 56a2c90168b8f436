"""Developer tool (GPU box): 400 engine.simulate() calls alternating tally and sampled-history modes: results stay put, memory does not grow."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pvtrace_amd import engine
from tests import scenes
scene = scenes.lsc_equivalent()
free0 = torch.cuda.mem_get_info()[0]
tic = time.time()
tot = 0
for k in range(400):
    r = engine.simulate(scene, 20000, seed=k, record_every=0 if k % 3 else 100)
    tot += int(r.data["rec_distinct"][7])
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print(f"400 simulate() calls (new Session each): {time.time()-tic:.2f} s, entering {tot}, device memory delta {(free0-free1)/2**20:.1f} MiB")
tic = time.time()
n = 0
for r, traced in engine.simulate_stream(scene, 2_000_000_000 // 10, bundle=4_000_000, seed=3, record_every=0):
    n += r.num_rays
print(f"streamed {n/1e6:.0f} M photons in {time.time()-tic:.2f} s -> {n/(time.time()-tic)/1e9:.2f} G photons/s end to end")
