"""Developer tool (GPU box): what moving the reference's host arrays costs on this box -- pageable against pinned against
registered (hipHostRegister) memory, both directions -- next to the host-buffer entry `pvt_trace_bundle` itself.
usage: python tools/gpu_host_io.py"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")


def best(fn, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append(time.perf_counter() - t)
    return min(out), sorted(out)[len(out) // 2]


for mb in (8, 56, 512):
    n = mb * (1 << 20) // 8
    src = np.random.rand(n)
    dst = torch.empty(n, dtype=torch.float64, device=dev)
    b, m = best(lambda: dst.copy_(torch.from_numpy(src)))
    print(f"H2D {mb:4d} MB pageable numpy -> device        best {b*1e3:7.2f} ms ({mb/1024/b:6.1f} GB/s)  median {m*1e3:7.2f} ms", flush=True)
    t = time.perf_counter(); pin = torch.empty(n, dtype=torch.float64, pin_memory=True); t_alloc = time.perf_counter() - t
    pin.numpy()[:] = src
    b, m = best(lambda: dst.copy_(pin, non_blocking=True))
    print(f"H2D {mb:4d} MB pinned -> device                best {b*1e3:7.2f} ms ({mb/1024/b:6.1f} GB/s)  (first pinned alloc {t_alloc*1e3:.1f} ms)", flush=True)
    b, m = best(lambda: (pin.numpy().__setitem__(slice(None), src), dst.copy_(pin, non_blocking=True)))
    print(f"H2D {mb:4d} MB numpy -> pinned (memcpy) -> device best {b*1e3:7.2f} ms ({mb/1024/b:6.1f} GB/s)", flush=True)
    # register the caller's own pages, copy, unregister
    ptr = src.ctypes.data
    def reg_copy():
        assert hip.hipHostRegister(C.c_void_p(ptr), C.c_size_t(src.nbytes), 0) == 0
        assert hip.hipMemcpy(C.c_void_p(dst.data_ptr()), C.c_void_p(ptr), C.c_size_t(src.nbytes), 1) == 0
        assert hip.hipHostUnregister(C.c_void_p(ptr)) == 0
    b, m = best(reg_copy)
    t = time.perf_counter(); hip.hipHostRegister(C.c_void_p(ptr), C.c_size_t(src.nbytes), 0); t_reg = time.perf_counter() - t
    b2, _ = best(lambda: hip.hipMemcpy(C.c_void_p(dst.data_ptr()), C.c_void_p(ptr), C.c_size_t(src.nbytes), 1))
    t = time.perf_counter(); hip.hipHostUnregister(C.c_void_p(ptr)); t_unreg = time.perf_counter() - t
    print(f"H2D {mb:4d} MB register + copy + unregister     best {b*1e3:7.2f} ms ({mb/1024/b:6.1f} GB/s)  [register {t_reg*1e3:.2f} ms, copy alone {b2*1e3:.2f} ms, unregister {t_unreg*1e3:.2f} ms]", flush=True)
    # D2H
    host = np.empty(n)
    b, m = best(lambda: torch.from_numpy(host).copy_(dst))
    print(f"D2H {mb:4d} MB device -> pageable numpy        best {b*1e3:7.2f} ms ({mb/1024/b:6.1f} GB/s)", flush=True)
    b, m = best(lambda: pin.copy_(dst, non_blocking=True))
    print(f"D2H {mb:4d} MB device -> pinned                best {b*1e3:7.2f} ms ({mb/1024/b:6.1f} GB/s)", flush=True)
    b, m = best(lambda: dst.cpu())
    print(f"D2H {mb:4d} MB tensor.cpu() (fresh pageable)   best {b*1e3:7.2f} ms ({mb/1024/b:6.1f} GB/s)", flush=True)
    del pin, dst

# the host-buffer entry itself
from benchmarks.configs import cfg2_lsc
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
scene = cfg2_lsc()
compiled = compile_scene(scene)
for n in (1_000_000, 4_000_000):
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=1)
    _kernel.trace_bundle(compiled, pos[:1000], dirs[:1000], wl[:1000], 1, 1000, 128, 0, 1, 0)
    times = []
    for rep in range(7):
        timing = {}
        t = time.perf_counter()
        _kernel.trace_bundle(compiled, pos, dirs, wl, 1 + rep, 1000, 128, 0, 1, 0, timing=timing)
        times.append((time.perf_counter() - t, timing["kernel_ms"]))
    b = min(times)
    print(f"pvt_trace_bundle host arrays n={n}: best {b[0]*1e3:.2f} ms ({n/b[0]/1e6:.0f} M photons/s), kernel {b[1]:.2f} ms; all: "
          + " ".join(f"{t*1e3:.2f}" for t, _ in times), flush=True)
