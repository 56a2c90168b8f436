"""bench.py's scenes are built with the product API alone (benchmarks/configs.py); they must flatten to the very
tables the parity tests and the golden fixtures use (tests/scenes.py), and the host-side helpers of the bench
must behave without a GPU."""
import os
import subprocess
import sys

import numpy as np

import bench
from benchmarks import configs
from pvtrace_amd.engine import compile_scene
from tests import scenes
from tests.util import assert_same_tables, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_scenes_flatten_to_the_tested_tables():
    for mine, theirs in ((configs.cfg2_lsc, scenes.lsc_equivalent), (configs.cfg4_nested_cylinders, scenes.nested_cylinders),
                         (configs.cfg5_coated_slab, scenes.coated_slab)):
        a, b = compile_scene(mine()), compile_scene(theirs())
        ta, tb = a.tables(), b.tables()
        assert ta.keys() == tb.keys()
        for key in ta:
            assert np.array_equal(np.asarray(ta[key]), np.asarray(tb[key])), (mine.__name__, key)
        assert a.recorder_names == b.recorder_names and a.node_names == b.node_names
    # the headline scene IS the scene of the 10^6-photon reference fixture; cfg4 likewise
    assert_same_tables(compile_scene(configs.cfg2_lsc()), load_golden("tallies_lsc_1e6.npz"))
    assert_same_tables(compile_scene(configs.cfg4_nested_cylinders()), load_golden("tallies_nested_cylinders_1e6.npz"))


def test_rank_pinning_partitions_the_allowed_cores():
    allowed = sorted(os.sched_getaffinity(0))
    try:
        if len(allowed) >= 2:
            world = 2
            slices = []
            for r in range(world):
                os.sched_setaffinity(0, allowed)
                slices.append(bench.pin_rank_to_cores(r, world))
            assert all(s for s in slices) and not set(slices[0]) & set(slices[1])
            assert set(slices[0]) | set(slices[1]) <= set(allowed)
        assert bench.pin_rank_to_cores(0, 1) is None
    finally:
        os.sched_setaffinity(0, allowed)
    assert 1 <= bench.usable_cores() <= len(allowed)


def test_bench_without_a_gpu_fails_loudly_with_an_error_line():
    """No CPU path: on a box without a GPU the bench must say so (and print the line the driver reads)."""
    import torch

    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=600)
    assert done.returncode != 0 and "no CPU path" in (done.stdout + done.stderr)
