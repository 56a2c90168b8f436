"""Developer diagnostic (run on the GPU box): per-scene GPU-vs-oracle diff with
verbose mismatch reports, plus a quick throughput probe.  Not part of the test
suite; tests/test_gpu_*.py are the real gates."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import oracle as O
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes


def diff(name, gpu, cpu):
    bad = []
    for key in cpu:
        a, b = np.asarray(gpu[key]), np.asarray(cpu[key])
        if key == "rec_sums":
            ok = np.allclose(a, b, rtol=1e-12, atol=0)
        else:
            ok = a.shape == b.shape and np.array_equal(a, b)
        if not ok:
            bad.append(key)
    if bad:
        print(f"  [{name}] MISMATCH in {bad}")
        for key in bad[:4]:
            a, b = np.asarray(gpu[key]), np.asarray(cpu[key])
            if a.shape != b.shape:
                print("    shape", a.shape, b.shape); continue
            idx = np.argwhere(a != b)
            print(f"    {key}: {len(idx)} differing; first {idx[:3].tolist()} gpu={a[tuple(idx[0])]} cpu={b[tuple(idx[0])]}")
    return not bad


def main():
    n = int(os.environ.get("N", "3000"))
    allok = True
    for name, mk in scenes.ALL_SCENES.items():
        sc = mk()
        c = compile_scene(sc)
        pos, d, wl, _ = emit_bundle(sc, n, seed=123)
        for rec_every, maxev, maxsteps, em in [(1, 64, 1000, 0), (0, 128, 50, 1), (7, 16, 1000, 2)]:
            gpu = _kernel.trace_bundle(c, pos, d, wl, 42, maxsteps, maxev, em, 1, rec_every)
            cpu = O.trace_bundle(c, pos, d, wl, 42, maxsteps, maxev, em, 1, rec_every, math_mode=1)
            ok = diff(f"{name} rec={rec_every}", gpu, cpu)
            allok &= ok
            print(f"{name:18s} rec_every={rec_every} max_events={maxev} maxsteps={maxsteps} emit={em}: {'OK' if ok else 'FAIL'}")
    print("ALL OK" if allok else "SOME FAILED")

    # throughput probe on the headline scene
    sc = scenes.lsc_equivalent()
    c = compile_scene(sc)
    for nrays in (100_000, 1_000_000, 4_000_000):
        pos, d, wl, _ = emit_bundle(sc, nrays, seed=5)
        for rep in range(3):
            t = {}
            tic = time.perf_counter()
            out = _kernel.trace_bundle(c, pos, d, wl, 1, 1000, 128, 0, 1, 0, timing=t)
            wall = time.perf_counter() - tic
        print(f"LSC n={nrays}: kernel {t['kernel_ms']:.3f} ms -> {nrays / t['kernel_ms'] / 1e3:.2f} M photons/s (wall {wall*1e3:.1f} ms);"
              f" top={out['rec_distinct'][0] / nrays:.5f} lost={out['rec_distinct'][6] / nrays:.5f}")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
