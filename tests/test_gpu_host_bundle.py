"""The host-buffer entry `pvt_trace_bundle` uploads the rays in chunks and traces a chunk while the next one crosses
PCIe (three trace streams used in turn, inner boundaries on multiples of record_every).  Which chunk carries a ray must
not show anywhere: the results are those of the unsplit call and of the CPU referee, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    for key in b:
        if key == "rec_sums":   # (f64 sums: the order of addition differs between splits, as between thread counts)
            assert np.allclose(a[key], b[key], rtol=1e-11, atol=0), (what, key)
        else:
            assert np.array_equal(a[key], b[key]), (what, key)


@pytest.mark.parametrize("name,every,max_events", [("lsc_equivalent", 0, 16), ("lsc_equivalent", 1, 24),
                                                   ("nested_cylinders", 7, 12), ("kitchen_sink", 1000, 40)])
def test_chunked_upload_is_invisible_in_the_results(monkeypatch, name, every, max_events):
    n, seed = 20_011, 33
    scene = scenes.ALL_SCENES[name]()
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=4)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 4, every, math_mode=O.MATH_PORTABLE)
    monkeypatch.setenv("PVT_HOST_CHUNK_RAYS", "0")          # one upload, one launch
    whole = _kernel.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 1, every)
    _same(whole, cpu, "unsplit")
    for chunk in (1500, 4096, 19_999):                      # 14, 5 and 2 chunks (rounded up to multiples of record_every)
        monkeypatch.setenv("PVT_HOST_CHUNK_RAYS", str(chunk))
        split = _kernel.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 1, every)
        _same(split, cpu, f"chunks of {chunk}")


def test_default_chunking_of_a_large_bundle_matches_the_unsplit_call(monkeypatch):
    """10^6 rays: two chunks by default (no switch set); integer tallies equal those of the one-launch call."""
    n = 1_000_000
    scene = scenes.ALL_SCENES["lsc_equivalent"]()
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=9)
    monkeypatch.delenv("PVT_HOST_CHUNK_RAYS", raising=False)
    timing = {}
    split = _kernel.trace_bundle(compiled, pos, dirs, wl, 5, 1000, 2, 0, 1, 0, timing=timing)
    monkeypatch.setenv("PVT_HOST_CHUNK_RAYS", "0")
    whole = _kernel.trace_bundle(compiled, pos, dirs, wl, 5, 1000, 2, 0, 1, 0)
    _same(split, whole, "default chunks")
    assert split["rec_distinct"].sum() > n // 2 and timing["kernel_ms"] > 0


def test_the_kept_device_block_can_be_released_and_is_not_needed(monkeypatch):
    """The host-buffer entry keeps the device block of a call for the next one (ABI v13); `pvt_release_cached_memory`
    frees it, PVT_NO_HOST_CACHE never keeps it: the same numbers every way, a smaller bundle after a larger one included
    (the kept block is larger than it needs: stale rays beyond its own must never be read)."""
    from pvtrace_amd.engine import native

    lib = native.load_library()
    scene = scenes.ALL_SCENES["lsc_equivalent"]()
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, 30_000, seed=2)
    big = _kernel.trace_bundle(compiled, pos, dirs, wl, 3, 1000, 16, 0, 1, 5)
    small = _kernel.trace_bundle(compiled, pos[:7001], dirs[:7001], wl[:7001], 3, 1000, 16, 0, 1, 5)   # in the block `big` left
    lib.pvt_release_cached_memory()
    lib.pvt_release_cached_memory()                                                                   # (nothing kept: a no-op)
    small_again = _kernel.trace_bundle(compiled, pos[:7001], dirs[:7001], wl[:7001], 3, 1000, 16, 0, 1, 5)
    monkeypatch.setenv("PVT_NO_HOST_CACHE", "1")
    big_again = _kernel.trace_bundle(compiled, pos, dirs, wl, 3, 1000, 16, 0, 1, 5)
    _same(small_again, small, "after release")
    _same(big_again, big, "without the cache")
    cpu = O.trace_bundle(compiled, pos[:7001], dirs[:7001], wl[:7001], 3, 1000, 16, 0, 4, 5, math_mode=O.MATH_PORTABLE)
    _same(small, cpu, "small bundle in a larger kept block")


def test_chunks_inside_the_shards_of_a_device_list(monkeypatch):
    """`pvt_trace_bundle_multi` gives every entry of the device list a shard, and every shard uploads in chunks of its own:
    shard edges and chunk edges on multiples of record_every, every ray still on the stream seed + global index."""
    n, seed, every, max_events = 20_011, 41, 7, 12
    scene = scenes.ALL_SCENES["nested_cylinders"]()
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=6)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 4, every, math_mode=O.MATH_PORTABLE)
    monkeypatch.setenv("PVT_HOST_CHUNK_RAYS", "1500")
    for devices in ([0, 0], [0, 0, 0]):
        got = _kernel.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 1, every, devices=devices)
        _same(got, cpu, f"devices {devices}, chunks of 1500")
