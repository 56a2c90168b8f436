"""The reference's own engine benchmark (benchmarks/benchmark_engine.py:26-134), run on the MI355X
engine: the synthetic 5 x 5 x 1 cm dye slab, 200 000 rays with full event histories
(`max_events=256`, `record_every=1`) and the recorder-only mode (2 000 000 rays, tallies for every
ray, paths for every 1000th).  There is no Python tracer and no CPU path to compare with here (the
reference prints its Python tracer's ~10^3 rays/s and the Cython kernel's 0.4-3 * 10^6 rays/s next
to it); `bench.py` reports the CPU port of the reference kernel beside the GPU for the headline
configuration.

    python benchmarks/benchmark_engine.py          # on a machine with an MI355X
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pvtrace_amd import engine                                   # noqa: E402
from pvtrace_amd.engine import Heatmap, Histogram, Recorder      # noqa: E402


def make_lsc_scene():
    """The reference harness's scene (benchmark_engine.py:26-55): Gaussian dye slab in a sphere,
    collimated light from below -- kept once, in the shared scene library."""
    from tests import scenes

    return scenes.bench_slab(recorders=False)


def best_of(fn, repeats=3):
    best = None
    for _ in range(repeats):
        tic = time.perf_counter()
        result = fn()
        wall = time.perf_counter() - tic
        if best is None or wall < best[0]:
            best = (wall, result)
    return best


def main():
    if not engine.is_available():
        print("HIP engine not built or no GPU visible; run: python -c 'import __graft_entry__ as g; g.build()'")
        return
    scene = make_lsc_scene()
    engine.simulate(scene, 1000, seed=0)    # load the library, warm the device

    n_engine = 200000
    wall, result = best_of(lambda: engine.simulate(scene, n_engine, seed=0, max_events=256))
    events = int(result.data["counts"].sum())
    print(f"engine (histories) {n_engine:>8d} rays  trace {result.elapsed * 1e3:8.2f} ms {n_engine / result.elapsed:>14,.0f} rays/s"
          f"  end to end {wall * 1e3:8.1f} ms {n_engine / wall:>13,.0f} rays/s  ({events} events)")

    # Recorder-only mode (benchmark_engine.py:97-130)
    slab = [n for n in scene.root.children if n.name == "slab"][0]
    slab.recorders = [
        Recorder("top-escape", event="escaping", facet=(0, 0, 1),
                 histograms=[Histogram("wavelength", 400, 900, 100),
                             Heatmap("x", "y", (-2.5, 2.5, 50), (-2.5, 2.5, 50))]),
        Recorder("lost", event="lost"),
    ]
    n_rec = 2000000
    wall, result = best_of(lambda: engine.simulate(scene, n_rec, seed=0, record_every=1000))
    top = result.recorders["top-escape"]
    print(f"recorders          {n_rec:>8d} rays  trace {result.elapsed * 1e3:8.2f} ms {n_rec / result.elapsed:>14,.0f} rays/s"
          f"  end to end {wall * 1e3:8.1f} ms {n_rec / wall:>13,.0f} rays/s  "
          f"(top-escape {top.rays}, {result.num_recorded} paths kept)")


if __name__ == "__main__":
    main()
