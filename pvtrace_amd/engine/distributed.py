"""Photons sharded across the GPUs of a node: one process per GPU, tallies summed
with a single RCCL all-reduce over xGMI.

The reference has no collective anywhere; its only "reduction"
is the per-thread -> global sum of the recorder accumulators
(pvtrace/engine/_kernel.pyx:1099-1102) and its streaming rule is "bundle b
traces rays with seeds seed + traced + i" (api.py:249-264).  Both map directly:

* rank r owns the contiguous index range [n*r/W, n*(r+1)/W); ray i always uses
  RNG stream ``seed + i`` and (device emission) emission stream
  ``(emit_seed + i)``, so the set of photon histories is identical for every
  world size;
* there is NO data-path exchange while tracing; after the kernel, the integer
  tallies (distinct | crossings | bins as one int64 buffer) and the f64 moment
  sums are all-reduced (sum).  Payload is O(recorders + bins) — tens of KB —
  so the collective is latency-bound (~tens of microseconds), independent of
  the photon count;
* sampled event logs stay on the rank that traced them; shard boundaries are multiples of
  `record_every`, so together they are the single-process log.

`backend="nccl"` is RCCL on ROCm.  The CPU tests run the sharding and the reduction
(`run_sharded`) with gloo around a trace of their own.
"""
import time

import numpy as np

from pvtrace_amd.engine import native
from pvtrace_amd.engine.compiler import EMIT_METHODS, compile_scene


def shard_range(num_rays, rank, world_size, align=1):
    """[start, stop) of the global ray indices owned by `rank`.  With `align` > 1 the inner
    boundaries are multiples of it: a job that keeps the history of every `record_every`-th ray
    shards on those multiples, so each rank samples exactly the rays a single process would
    (the kernel samples by index within its bundle, `_kernel.pyx:1085-1089`) and the union of
    the shards' event logs IS the single-process log."""
    align = max(int(align), 1)

    def edge(r):
        if r <= 0:
            return 0
        if r >= world_size:
            return num_rays
        return min(((num_rays * r) // world_size) // align * align, num_rays)

    return edge(rank), edge(rank + 1)


def all_reduce_tallies(tallies, group=None):
    """Sum recorder accumulators over all ranks, in place (2 collectives: one
    int64 buffer holding distinct | crossings | bins, one f64 buffer of moment sums)."""
    import torch
    import torch.distributed as dist

    if "_ints" in tallies:  # DeviceScene.new_tallies(): the tables are views of two buffers
        dist.all_reduce(tallies["_ints"], op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tallies["_sums"], op=dist.ReduceOp.SUM, group=group)
        return tallies
    nrec = tallies["rec_distinct"].numel()
    ints = torch.cat([tallies["rec_distinct"], tallies["rec_crossings"], tallies["rec_bins"]])
    dist.all_reduce(ints, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(tallies["rec_sums"], op=dist.ReduceOp.SUM, group=group)
    tallies["rec_distinct"].copy_(ints[:nrec])
    tallies["rec_crossings"].copy_(ints[nrec:2 * nrec])
    tallies["rec_bins"].copy_(ints[2 * nrec:])
    return tallies


def run_sharded(scene, num_rays, max_events, record_every, group, trace_shard, prepare=None):
    """What every rank of a sharded job does around its own trace: take the index range of the rank, trace it
    (`trace_shard(compiled, start, stop) -> (tallies, finish)`: `tallies` holds torch tensors -- on whatever device
    the group's backend reduces -- and `finish(tallies) -> data` turns the reduced tallies and the rank's event log into
    the reference's `data` dict), sum the tallies over the ranks, wrap the result.  `simulate_sharded` passes the
    GPU trace; anything that produces the same arrays shards and reduces the same way.
    `prepare(compiled, start, stop)` (optional) runs BEFORE the clock starts: whatever the trace needs resident first --
    tables uploaded, buffers allocated, device idle -- so that `EngineResult.elapsed` wraps the trace and the
    reduction only, as `simulate()`'s does (reference convention, api.py:232-245)."""
    import torch.distributed as dist

    from pvtrace_amd.engine import emit as emit_mod
    from pvtrace_amd.engine.api import EngineResult

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    start, stop = shard_range(num_rays, rank, world, align=record_every)
    compiled = compile_scene(scene)
    sources = emit_mod.sources_for(scene, num_rays)[start:stop]
    if prepare is not None:
        prepare(compiled, start, stop)
    tic = time.perf_counter()
    tallies, finish = trace_shard(compiled, start, stop)
    all_reduce_tallies(tallies, group=group)
    data = finish(tallies)
    result = EngineResult(compiled, data, sources, max_events, record_every, time.perf_counter() - tic)
    result.shard = (start, stop)
    return result


def simulate_sharded(scene, num_rays, seed, emit_seed=0, maxsteps=1000, max_events=128,
                     emit_method="kT", record_every=0, device=None, group=None, rays=None):
    """Trace this rank's shard of a `num_rays` job and all-reduce the tallies.

    Must be called by every rank of an initialised torch.distributed group.
    Rays come from device-side emission (shard-invariant by construction) unless
    `rays` = (positions, directions, wavelengths) host arrays of the WHOLE job
    are given, in which case each rank slices its range.

    Returns an `EngineResult` whose recorder tallies are GLOBAL and whose event
    log (if any) covers the local shard; `.shard` = (start, stop).
    """
    import torch

    from pvtrace_amd.engine import emit as emit_mod
    from pvtrace_amd.engine.api import _default_device, download

    if emit_method not in EMIT_METHODS:
        raise ValueError(f"emit_method must be one of {sorted(EMIT_METHODS)}")
    if device is None:
        device = _default_device()
    opened = []
    ready = {}

    def prepare(compiled, start, stop):
        # everything but the trace: emitter tables, the scene's upload (with its grid / BVH planning), rays, buffers
        n_local = stop - start
        emitter = emit_mod.EmitterTables(scene, strict=True) if rays is None else None
        dscene = native.DeviceScene(compiled, device=device, emitter=emitter)
        opened.append(dscene)
        dev = torch.device("cuda", device)
        dev_rays = None
        if rays is not None:
            dev_rays = tuple(torch.from_numpy(np.ascontiguousarray(np.asarray(a)[start:stop])).to(dev) for a in rays)
        tallies = dscene.new_tallies()
        log = (dscene.new_event_log(n_local, record_every, max_events) if record_every > 0 and n_local > 0 else None)
        ready.update(dscene=dscene, dev_rays=dev_rays, tallies=tallies, log=log)
        torch.cuda.synchronize(device)

    def trace_shard(compiled, start, stop):
        n_local = stop - start
        dscene, dev_rays, tallies, log = ready["dscene"], ready["dev_rays"], ready["tallies"], ready["log"]
        if n_local > 0:
            dscene.trace(dev_rays, n_local, int(seed), tallies, log=log, ray_offset=start,
                         emit_seed=int(emit_seed), record_every=int(record_every),
                         maxsteps=int(maxsteps), max_events=int(max_events),
                         emit_method=EMIT_METHODS[emit_method])

        def finish(reduced):
            torch.cuda.synchronize(device)
            return download(compiled, reduced, log, n_local, record_every, max_events)

        return tallies, finish

    try:
        with torch.cuda.device(device):
            return run_sharded(scene, num_rays, max_events, record_every, group, trace_shard, prepare=prepare)
    finally:
        for dscene in opened:
            dscene.close()
