"""Bundle pipelining: keep several photon bundles in flight on separate HIP streams.

A launch of n photons ends with a *drain*: once the ray cursor is dry the kernel lives
for `longest remaining history x step latency` (~0.4 ms for the LSC scene whatever n is)
while most of the chip idles.  For one huge bundle that is noise; for a STREAM of
moderate bundles (the reference's `simulate_stream`, api.py:249-264; the studio's
consumer loop, studio/server.py:218-256) it is half the wall time at 10^6 photons per
bundle.  `BundlePipeline` enqueues bundle k+1 on another stream before bundle k has
finished, so the next bundle's bulk phase fills the CUs the draining bundle vacates.
Every bundle is still traced completely and tallied into its own buffers; results are
identical to running the bundles one after another (per-ray RNG streams).
"""
from pvtrace_amd.engine import native


class BundlePipeline:
    def __init__(self, dscene, depth=2, distributed=False, group=None, reduce="end", carry=True):
        """`distributed`: the bundles of this pipeline are one rank's shard of a multi-GPU job.
        `reduce="end"` sums the tallies over ranks ONCE, when the totals are asked for (the sum
        over bundles commutes with the sum over ranks, so one RCCL all-reduce of a few KB serves
        the whole job); `reduce="bundle"` all-reduces every bundle as it completes, for consumers
        that show global running totals.
        `carry`: the bundles are parts of ONE job whose totals are wanted (every mode but per-bundle all-reduces):
        a launch does not trace its last, longest histories to completion but hands the photons still alive to the
        next launch on its stream (PVT_FLAG_CARRY_OUT); `reduce_totals()` / `reset_totals()` finish what is
        waiting, so totals always cover every photon submitted, completely traced."""
        import os

        import torch

        if reduce not in ("end", "bundle"):
            raise ValueError("reduce must be 'end' or 'bundle'")
        self.reduce = reduce
        self._reduced = False
        self._unordered = set()   # streams whose totals were zero-filled on streams[0] by the last reduce_totals()
        self.carry = bool(carry) and not (distributed and reduce == "bundle")
        if os.environ.get("PVT_NO_CARRY"):   # developer A/B switch
            self.carry = False
        self._parked = {}         # stream index -> (maxsteps, max_events, emit_method) of the launch that parked photons

        self.torch = torch
        self.dscene = dscene
        self.device = torch.device("cuda", dscene.device)
        self.depth = max(1, int(depth))
        self.distributed = distributed
        self.group = group
        # launches that overlap run best with fewer persistent workgroups each (measured on the
        # LSC, 10^6-photon bundles, photons carried between launches: 2 per CU with two or three in flight, 4 alone;
        # without carrying -- launches that drain -- 3 per CU with two in flight)
        self.workgroups_per_cu = {1: 4}.get(self.depth, 2)
        if not self.carry and self.depth == 2:
            self.workgroups_per_cu = 3
        if os.environ.get("PVT_PIPE_WGS"):   # developer sweep
            self.workgroups_per_cu = int(os.environ["PVT_PIPE_WGS"])
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        # torch hands out streams from a small pool and a resident scene outlives a pipeline: photons an ABANDONED
        # pipeline left parked on one of these stream handles (an exception between two bundles, an object simply
        # dropped) belong to a dead job and must not be resumed into this one -- while a handle that a LIVE pipeline
        # on this scene still uses must not be shared at all (DeviceScene.claim_stream keeps the ownership)
        self._claimed = []
        try:
            for s_ in self.streams:
                dscene.claim_stream(self, s_.cuda_stream)
                self._claimed.append(s_.cuda_stream)
        except Exception:
            for h in self._claimed:
                dscene.release_stream(self, h)
            self._claimed = []
            raise
        self.slots = [dscene.new_tallies() for _ in range(self.depth)]
        # The kernel ADDS its tallies with atomics (one per non-zero slot per workgroup), so the launches of every
        # stream can add into ONE running total: nothing to fold when the totals are read (the fold was eight tiny
        # kernels and two cross-stream waits at the end of every job).  Per-bundle all-reduces add each bundle's own
        # numbers with a plain torch `+=` on the bundle's stream, which is not atomic: they keep a total per stream.
        if distributed and reduce == "bundle":
            self.totals = [dscene.new_tallies() for _ in range(self.depth)]
        else:
            self.totals = [dscene.new_tallies()] * self.depth
        self.events = []       # (start, stop) HIP events of every trace launch
        self.submitted = 0
        self.wait_for_inputs()   # the zero-fills above ran on the current stream

    def submit(self, rays, n_rays, seed, ray_offset=0, emit_seed=0, maxsteps=1000, max_events=128,
               emit_method=0, timed=True, tail=False, closing=False):
        """Enqueue one tally-mode bundle (record_every=0) on the next stream; returns its slot.
        `tail`: nothing will follow this bundle soon (the last one of a job): it is launched at the full width
        of a lone launch, so that its own end does not run at the reduced width chosen for overlapping bundles.
        `closing`: no further bundle will follow on this bundle's stream before the totals are read (the last
        `depth` bundles of a job): it finishes its own photons and the ones it resumed instead of parking them for a
        launch that would carry no new rays."""
        torch = self.torch
        if self._reduced and self.distributed and self.reduce == "end":
            raise RuntimeError("totals already reduced over the ranks; call reset_totals() before submitting more bundles")
        k = self.submitted % self.depth
        if k in self._parked and self._parked[k] != (maxsteps, max_events, emit_method):
            raise ValueError(
                f"bundle traced with (maxsteps, max_events, emit_method) = {(maxsteps, max_events, emit_method)} while photons "
                f"parked on its stream were traced with {self._parked[k]}: the bundles of a pipeline are parts of one job; "
                "call reduce_totals() / reset_totals() (they finish what is parked) before changing the rules")
        self.submitted += 1
        stream, tallies, total = self.streams[k], self.slots[k], self.totals[k]
        if k in self._unordered:
            # reduce_totals() zero-filled totals[k] on streams[0]: this launch adds into that buffer from
            # another stream and must be ordered after the zero-fill, or its tallies could be wiped
            stream.wait_stream(self.streams[0])
            self._unordered.discard(k)
        self._reduced = False
        # The kernel ADDS its tallies to the buffers it is given (atomics, once per workgroup), so a bundle can be
        # traced straight into its stream's running totals: no zero-fill and no accumulate kernels per bundle.
        # Only per-bundle all-reduces need the bundle's own numbers.
        per_bundle = self.distributed and self.reduce == "bundle"
        with torch.cuda.stream(stream):
            if per_bundle:
                tallies["_ints"].zero_()
                tallies["_sums"].zero_()
            ev = None
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(stream)
            self.dscene.trace(rays, n_rays, seed=seed, tallies=tallies if per_bundle else total, ray_offset=ray_offset,
                              emit_seed=emit_seed, record_every=0, maxsteps=maxsteps,
                              max_events=max_events, emit_method=emit_method,
                              stream=stream.cuda_stream, workgroups_per_cu=4 if tail else self.workgroups_per_cu,
                              carry_out=self.carry and not (tail or closing))
            if self.carry:
                if tail or closing:
                    self._parked.pop(k, None)   # a tail launch finishes what it resumed
                else:
                    self._parked[k] = (maxsteps, max_events, emit_method)
            if timed:
                ev[1].record(stream)
                self.events.append(ev)
            if per_bundle:
                from pvtrace_amd.engine.distributed import all_reduce_tallies

                all_reduce_tallies(tallies, group=self.group)
                total["_ints"] += tallies["_ints"]
                total["_sums"] += tallies["_sums"]
        return k

    def wait_for_inputs(self):
        """Make every pipeline stream wait for work already queued on the current stream
        (e.g. the upload of the rays)."""
        cur = self.torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def finish_parked(self):
        """Trace the photons parked on the pipeline's streams to completion (one launch without new rays per
        stream that holds any), into that stream's totals."""
        torch = self.torch
        for k, (maxsteps, max_events, emit_method) in sorted(self._parked.items()):
            stream = self.streams[k]
            if k in self._unordered:
                stream.wait_stream(self.streams[0])
                self._unordered.discard(k)
            with torch.cuda.stream(stream):
                self.dscene.trace(None, 0, seed=0, tallies=self.totals[k], record_every=0, maxsteps=maxsteps,
                                  max_events=max_events, emit_method=emit_method, stream=stream.cuda_stream,
                                  workgroups_per_cu=4, carry_out=False)
        self._parked = {}

    def close(self):
        """Drop what an unfinished job left parked on the pipeline's streams (nothing is traced)."""
        for k in list(self._parked):
            try:
                self.dscene.carry_discard(self.streams[k].cuda_stream)
            except Exception:   # noqa: BLE001 -- the scene may be gone already
                pass
        self._parked = {}
        for h in getattr(self, "_claimed", []):
            try:
                self.dscene.release_stream(self, h)
            except Exception:   # noqa: BLE001
                pass
        self._claimed = []

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def reset_totals(self):
        self.finish_parked()      # photons of earlier bundles must not be tallied into what follows
        self.synchronize()
        for k, t in enumerate(self.totals):
            if k == 0 or t is not self.totals[0]:
                t["_ints"].zero_()
                t["_sums"].zero_()
        self.events = []
        self._reduced = False
        self._unordered = set()
        self.wait_for_inputs()   # the zero-fills ran on the current stream

    def reduce_totals(self):
        """Fold the per-stream totals into one buffer and, for a distributed job in
        `reduce="end"` mode, sum it over the ranks (two small RCCL all-reduces for the whole
        job).  Enqueued on the first pipeline stream; idempotent until the next submit/reset."""
        torch = self.torch
        if self._reduced:
            return
        self.finish_parked()
        first = self.streams[0]
        for s in self.streams[1:]:
            first.wait_stream(s)
        with torch.cuda.stream(first):
            folded = False
            for t in self.totals[1:]:
                if t is self.totals[0]:
                    continue
                self.totals[0]["_ints"] += t["_ints"]
                self.totals[0]["_sums"] += t["_sums"]
                t["_ints"].zero_()
                t["_sums"].zero_()
                folded = True
            if folded:
                self._unordered = set(range(1, self.depth))
            if self.distributed and self.reduce == "end":
                from pvtrace_amd.engine.distributed import all_reduce_tallies

                all_reduce_tallies(self.totals[0], group=self.group)
        self._reduced = True

    def totals_host(self):
        """Sum of every bundle submitted since the last reset -> host numpy dict."""
        self.reduce_totals()
        self.synchronize()
        ints, sums = self.totals[0]["_ints"], self.totals[0]["_sums"]
        c = self.dscene.compiled
        nrec = max(int(c.rec_node.shape[0]), 1)
        ints, sums = ints.cpu().numpy(), sums.cpu().numpy()
        r = int(c.rec_node.shape[0])
        return {
            "rec_distinct": ints[:nrec][:r], "rec_crossings": ints[nrec:2 * nrec][:r],
            "rec_bins": ints[2 * nrec:][: int(c.total_bins)],
            "rec_sums": sums[: r * 8].reshape(r, 4, 2),
        }

    def kernel_ms(self):
        self.synchronize()
        return [a.elapsed_time(b) for a, b in self.events]


def trace_stream(scene, num_rays, bundle, seed, emit_seed=0, maxsteps=1000, max_events=128,
                 emit_method="kT", device=None, depth=2):
    """Tally-mode trace of `num_rays` photons as pipelined bundles with device-side emission.
    Returns (data dict with the rec_* arrays of the WHOLE job, elapsed seconds)."""
    import time

    import torch

    from pvtrace_amd.engine import emit as emit_mod
    from pvtrace_amd.engine.api import _default_device
    from pvtrace_amd.engine.compiler import EMIT_METHODS, compile_scene

    if emit_method not in EMIT_METHODS:
        raise ValueError(f"emit_method must be one of {sorted(EMIT_METHODS)}")
    compiled = compile_scene(scene)
    if device is None:
        device = _default_device()
    emitter = emit_mod.EmitterTables(scene, strict=True)
    dscene = native.DeviceScene(compiled, device=device, emitter=emitter)
    try:
        with torch.cuda.device(device):
            pipe = BundlePipeline(dscene, depth=depth)
            torch.cuda.synchronize(device)
            tic = time.perf_counter()
            traced = 0
            while traced < num_rays:
                n = min(bundle, num_rays - traced)
                pipe.submit(None, n, seed=int(seed), ray_offset=traced, emit_seed=int(emit_seed),
                            maxsteps=maxsteps, max_events=max_events,
                            emit_method=EMIT_METHODS[emit_method], timed=False, tail=traced + n >= num_rays,
                            closing=traced + n * depth >= num_rays)
                traced += n
            data = pipe.totals_host()
            elapsed = time.perf_counter() - tic
    finally:
        dscene.close()
    return compiled, data, elapsed
