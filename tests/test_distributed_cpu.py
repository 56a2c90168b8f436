"""N>1 path on CPU: two gloo processes shard a job by index range, trace their
shards (the oracle stands in for the kernel — there is no GPU here; the sharding and the
reduction are the product's `run_sharded`), all-reduce the tallies, and must reproduce the single-process result exactly for every integer
tally (f64 moment sums to rounding)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from pvtrace_amd.engine.distributed import run_sharded, shard_range
    from pvtrace_amd.engine.emit import emit_bundle
    from tests import scenes

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene = scenes.bench_slab(recorders=True)
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=55)

    def trace_shard(compiled, start, stop):
        # the product's sharding + reduction around a CPU trace (there is no GPU here): the oracle stands in for
        # the kernel, with the rank's index range and ray offset
        data = O.trace_bundle(compiled, pos[start:stop], dirs[start:stop], wl[start:stop], 77, 1000, 64, 0, 1, 10,
                              ray_offset=start, math_mode=O.MATH_PORTABLE)
        tallies = {k: torch.from_numpy(np.ascontiguousarray(data[k]).reshape(-1).copy())
                   for k in ("rec_distinct", "rec_crossings", "rec_sums", "rec_bins")}

        def finish(reduced):
            out = dict(data)
            for k in ("rec_distinct", "rec_crossings", "rec_bins"):
                out[k] = reduced[k].numpy()
            out["rec_sums"] = reduced["rec_sums"].numpy().reshape(data["rec_sums"].shape)
            return out

        return tallies, finish

    result = run_sharded(scene, n, 64, 10, None, trace_shard)
    assert result.shard == shard_range(n, rank, world, align=10)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), counts=result.data["counts"],
             **{k: result.data[k] for k in ("rec_distinct", "rec_crossings", "rec_sums", "rec_bins")})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition_the_job():
    from pvtrace_amd.engine.distributed import shard_range

    for n in (0, 1, 7, 1000, 10 ** 8 + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
            for align in (7, 1000):
                edges = [shard_range(n, r, world, align=align) for r in range(world)]
                assert edges[0][0] == 0 and edges[-1][1] == n
                assert all(a[1] == b[0] and a[1] % align == 0 for a, b in zip(edges, edges[1:]))


@pytest.mark.timeout(1200)
def test_two_rank_gloo_job_equals_single_process(tmp_path):
    import torch.multiprocessing as mp

    from oracle import oracle as O
    from pvtrace_amd.engine import compile_scene
    from pvtrace_amd.engine.emit import emit_bundle
    from tests import scenes

    n, world = 3001, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    scene = scenes.bench_slab(recorders=True)
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=55)
    single = O.trace_bundle(compiled, pos, dirs, wl, 77, 1000, 64, 0, 1, 10, math_mode=O.MATH_PORTABLE)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in ranks:  # every rank holds the GLOBAL tallies after the all-reduce
        for key in ("rec_distinct", "rec_crossings", "rec_bins"):
            assert np.array_equal(r[key], single[key]), key
        assert np.allclose(r["rec_sums"], single["rec_sums"], rtol=1e-12)
    assert single["rec_distinct"].sum() > 0
    # sampled histories stay with the rank that traced them: together they cover the job
    assert sum(len(r["counts"]) for r in ranks) >= len(single["counts"])
