"""Developer tool (GPU box): LDS bytes and grid of a tally launch, per test scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import scenes
from pvtrace_amd.engine import compile_scene, native
from pvtrace_amd.engine.emit import emit_bundle

for name in sys.argv[1:] or ["lsc_equivalent", "bench_slab", "coated_slab", "kitchen_sink", "hello_world", "nested_cylinders"]:
    sc = scenes.ALL_SCENES[name]()
    ds = native.DeviceScene(compile_scene(sc))
    pos, d, wl, _ = emit_bundle(sc, 100000, seed=1)
    dev = torch.device("cuda", 0)
    rays = [torch.as_tensor(x, device=dev).contiguous() for x in (pos, d, wl)]
    tallies = ds.new_tallies()
    ds.trace(tuple(rays), len(wl), 1, tallies, maxsteps=100)
    torch.cuda.synchronize()
    print(name, ds.launch_info(), flush=True)
