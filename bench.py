"""Headline benchmark: photons/s on the 5x5x1 cm Lumogen-F-Red LSC (BASELINE.json
configs[1]), one process per GPU.

    python bench.py                          # 1 GPU, default steps
    python bench.py --gpus N                 # re-executes itself under torch.distributed.run with N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: 10^6 photons PER GPU traced
through the scene with the recorder set of SURVEY.md §8(d) (escaping x 6 facets
with 80-bin wavelength histograms, lost, entering, reflected, killed),
`record_every=0`, `emit_method="kT"`, `maxsteps=1000`; every step uses fresh RNG
streams.  The initial rays (positions, directions, wavelengths — exactly the three
arrays the reference hands its kernel, pvtrace/engine/_kernel.pyx:903-914) are
emitted before the timed region and are resident in HBM, like the reference's own
convention (api.py:230-245 times only the trace).  With N>1 each rank traces its
own 10^6-photon index range per step (weak scaling) and the tallies are summed
with an RCCL all-reduce inside the timed region.

Rank 0 prints ONE JSON line; see the task contract for the fields.  `value` is the
MEDIAN of `1 + --repeats` independent, identically fenced windows of --steps steps
(`repeats` holds min / max / the first window).  `roofline` is for the trace kernel.
Its operative bound is FP64 VALU issue (`bound`, `frac` = VALU busy while the launches
overlap); the HBM figure the metric asks for sits beside it in `roofline.hbm`:
achieved = 56 algorithmic bytes/photon x photons per launch / mean launch duration
(HIP events on the launch stream) against 8 TB/s, with the PMC-measured traffic.
`roofline.steps_per_photon`, `roofline.live_lane_fraction` and the shader clock in the VALU-busy figure are counted /
read by the kernel itself during the timed windows (pvt_scene_counters, pvt_scene_clock), not taken from a profile;
`instruction_side.stale` says whether the committed PMC pass was sampled on the device code loaded now.  `configs` holds the same
measurement for BASELINE configs[3] (nested_cylinders) and configs[4] (coated slab +
scatterer) at 10^7 photons per GPU, pipelined, device-side emission.
`cpu_baseline` times the CPU referee (a port of the reference kernel, proven
bit-identical to it) on this box's host cores on a bounded sample.
"""
import argparse
import datetime
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PHOTONS_PER_GPU = 1_000_000
ALGORITHMIC_BYTES_PER_PHOTON = 56  # pos 24 + dir 24 + wavelength 8, read once
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s HBM3E


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the
    cgroup CPU quota (the GPU box exposes 256 logical CPUs but grants 16)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def pin_rank_to_cores(local_rank, local_world):
    """N ranks share the granted host cores: give each rank its own slice of the affinity mask (a submit loop is
    one busy Python thread; eight of them must not migrate over each other).  Returns the slice."""
    if not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return None
    allowed = sorted(os.sched_getaffinity(0))
    per = max(1, len(allowed) // local_world)
    mine = allowed[(local_rank * per) % len(allowed):][:per] or allowed
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def cpu_baseline(compiled, pos, dirs, wl, budget_s=(9.0, 6.0)):
    """Time the CPU referee (kind 'port': oracle/pvt_oracle.c, built -O3 -fopenmp like the reference's
    own kernel, pvtrace/engine/build.py:34) on a bounded sample of the same workload, with all usable
    host cores and with ONE thread, as the reference's harness does (benchmarks/benchmark_engine.py:121-124):
    repeated bundles (fresh RNG streams each) until about `budget_s` seconds have been spent per leg."""
    from oracle import oracle as O

    cores = usable_cores()
    n = pos.shape[0]
    O.trace_bundle(compiled, pos[:20000], dirs[:20000], wl[:20000], 1, 1000, 128, 0, cores, 0)  # warm

    def leg(threads, m, budget):
        photons, bundles = 0, 0
        tic = time.perf_counter()
        while True:
            O.trace_bundle(compiled, pos[:m], dirs[:m], wl[:m], 1 + bundles * m, 1000, 128, 0, threads, 0)
            photons += m
            bundles += 1
            elapsed = time.perf_counter() - tic
            if elapsed >= budget or bundles >= 400:
                return photons / elapsed, bundles, elapsed

    v_all, b_all, t_all = leg(cores, n, budget_s[0])
    m1 = min(n, 250_000)
    v_one, b_one, t_one = leg(1, m1, budget_s[1])
    return {
        "value": v_all, "unit": "photons/s", "cores": cores, "kind": "port", "value_1thread": v_one,
        "sample": f"{b_all} bundles x {n} photons of the same workload on {cores} OpenMP threads ({t_all:.2f} s), then "
                  f"{b_one} bundles x {m1} photons on 1 thread ({t_one:.2f} s); oracle/pvt_oracle.c, libm mode, "
                  f"tally mode, gcc -O3 -fopenmp",
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--photons", type=int, default=PHOTONS_PER_GPU, help="photons per GPU per step")
    ap.add_argument("--streams", type=int, default=3,
                    help="bundles kept in flight (HIP streams); 1 = strictly serial launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spinup-s", type=float, default=0.3,
                    help="untimed seconds of the same workload before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--ray-buffers", type=int, default=7,
                    help="distinct resident ray sets the steps rotate through (7 x 56 MB = 392 MB: past the "
                         "256 MiB Infinity Cache, so the ray stream of a step comes from HBM)")
    ap.add_argument("--repeats", type=int, default=15,
                    help="extra, independent timed windows of --steps steps after the first one; `value` is the "
                         "median of all of them")
    ap.add_argument("--sustained-s", type=float, default=12.5,
                    help="length of the sustained leg (back-to-back bundles) in seconds of GPU work (sized from the "
                         "sustained step time of a short probe, so that it really lasts this long: >= two ticks of a "
                         "5-s utilisation sampler); 0 = skip")
    ap.add_argument("--total-photons", type=int, default=100_000_000,
                    help="strong-scaling leg (BASELINE configs[2]): ONE job of this many photons split over the "
                         "ranks by index range, tallies all-reduced once; 0 = skip")
    ap.add_argument("--reduce", choices=("end", "bundle"), default="end",
                    help="multi-GPU: all-reduce the tallies once per job, inside the timed region "
                         "(default), or after every bundle")
    ap.add_argument("--config", default="cfg2",
                    help="scene of the main loop (developer flag for profiles; the contract's metric is cfg2). "
                         "cfg4/cfg5 use device-side emission")
    ap.add_argument("--scene-sizes", default="1,3,6,11",
                    help="tile-array sizes k of the scene-size leg (k x k slabs of the headline LSC in one world, k*k + 1 "
                         "nodes; benchmarks/configs.py tiles<k>); 'none' = skip")
    ap.add_argument("--extra-configs", default="cfg4,cfg5",
                    help="comma list of further configs timed after the cfg2 legs ('none' = skip)")
    ap.add_argument("--config-photons", type=int, default=10_000_000,
                    help="photons per GPU per window of an extra config (BASELINE: 10^7)")
    ap.add_argument("--config-sustained-s", type=float, default=2.0,
                    help="seconds of back-to-back bundles per extra config (its steady-state rate, without the closing "
                         "drain of a fenced window); 0 = skip")
    ap.add_argument("--rccl-timeout-s", type=float, default=180.0)
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU
    (the driver's own launch line, with a free port on 127.0.0.1)."""
    import __graft_entry__ as entry
    from pvtrace_amd.engine import native

    if not native.library_built():
        entry.build()   # once, before N ranks would race to build it
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


_CODE_HASH = []


def built_code_hash():
    """sha256 (16 hex digits) of the .text section of the gfx950 code object in the built library -- what
    tools/code_hash.sh prints; None when the LLVM binutils are not there."""
    if not _CODE_HASH:
        try:
            import __graft_entry__ as entry

            _CODE_HASH.append(entry.code_hash())
        except Exception:   # noqa: BLE001
            _CODE_HASH.append(None)
    return _CODE_HASH[0]


def live_counters(counters, photons):
    """What the kernel counted itself during the timed windows (pvt_scene_counters: two scalar adds per wave and trip of
    the photon loop, always on): trips per photon, the reference's loop count per photon, lanes holding a live photon."""
    if not counters or not counters.get("wave_iterations") or photons <= 0:
        return None
    return {
        "measured_in_this_run": True,
        "photons": photons,
        "steps_per_photon": counters["steps"] / photons,                        # = `count` of _kernel.pyx:655 per photon
        "lane_steps_per_photon": counters["lane_steps"] / photons,              # ... of which run (the rest: fused exits)
        "fused_exits_per_photon": counters["fused_exits"] / photons,
        "wave_iterations_per_photon": counters["wave_iterations"] / photons,
        "live_lane_fraction": counters["lane_utilisation"],                     # lanes holding a live photon / 64 when a wave steps
    }


def load_pmc(name, value_per_gpu, cus, live=None, clock=None):
    """Instruction-side numbers of a config.  Two kinds, labelled as such: what the kernel counts itself in THIS run
    (`in_kernel`: trips of the photon loop, live lanes per trip) and what only hardware counters can tell (vector
    instructions issued: the last committed PMC passes of this same command, tools/gpu_pmc.sh ->
    profiles/pmc_summary[_cfgN].json).  The two are tied together where possible: the PMC pass recorded the vector
    instructions PER TRIP of a wave; multiplied by the trips per photon counted in this run that is the instruction
    count per photon of THIS run's kernel as far as its control flow goes (a change that adds trips shows up; one that
    adds instructions inside a trip needs a new PMC pass)."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json" if name == "cfg2" else f"pmc_summary_{name}.json")
    if not os.path.exists(path):
        return None, ({"measured_in_this_run": False, "in_kernel": live, "built_kernel_text_hash": built_code_hash()} if live else None)
    try:
        summary = json.load(open(path))
        derived = summary.get("derived", {})
        built, sampled = built_code_hash(), summary.get("kernel_text_hash")
        side = {
            # the device code the counters were sampled on against the device code of the library loaded now (sha256 of the
            # gfx950 .text section, tools/code_hash.sh): `stale` = the kernels have changed since the PMC pass
            "pmc_kernel_text_hash": sampled, "built_kernel_text_hash": built,
            "stale": (sampled != built) if (sampled and built) else None,
            # counters come from committed rocprofv3 passes of this same command (rocprofv3 serialises dispatches while
            # it samples, so they cannot be taken in the timed run); only the photon rate they are scaled by is measured here
            "measured_in_this_run": False,
            "source": os.path.relpath(path, ROOT) + " (" + str(summary.get("stage", "")) + ")",
            "in_kernel": live,
            "valu_wave_instructions_per_photon": derived.get("valu_wave_instructions_per_photon"),
            "valu_lane_utilisation": derived.get("valu_lane_utilisation"),
            "wait_fraction_of_wave_cycles": derived.get("wait_any_fraction_of_wave_cycles"),
            # measured with counters, for ONE launch of this shape alone (rocprofv3 serialises
            # dispatches while sampling): 4*SQ_ACTIVE_INST_VALU / (SIMDs * GRBM_GUI_ACTIVE / 8)
            "valu_busy_measured_single_launch": derived.get("valu_busy_measured"),
        }
        per_photon = pmc_per_photon = derived.get("valu_wave_instructions_per_photon")
        per_trip = derived.get("valu_wave_instructions_per_wave_iteration")
        if per_trip and live:
            # The PMC passes run one launch at a time (rocprofv3 serialises dispatches), so their launches drain and take more
            # trips per photon than the overlapped, carried stream timed here; a trip at a few live lanes also issues fewer
            # instructions than a full one.  Scaled by the trips counted in THIS run the count is therefore a LOWER bound of
            # this run's, the PMC count itself an upper bound; the roofline fraction uses the lower one.
            per_photon = per_trip * live["wave_iterations_per_photon"]
            side["valu_wave_instructions_per_wave_iteration_pmc"] = per_trip
            side["valu_wave_instructions_per_photon_pmc"] = pmc_per_photon
            side["valu_wave_instructions_per_photon"] = per_photon
            side["valu_wave_instructions_per_photon_is"] = ("PMC instructions per trip x trips per photon counted in this run "
                                                            "(lower bound; the PMC run's own count is the upper bound)")
        if per_photon:
            # the operative ceiling: VALU issue.  Nominal: one wave64 FP64 instruction per SIMD every 4
            # cycles at 2.4 GHz.  ACHIEVABLE on this part with the kernel's four waves per SIMD: a pure
            # chain of v_fma_f64 issues one per 4.83-5.26 nominal cycles (tools/gpu_fma_peak.hip,
            # profiles/r02_fma_peak.txt) -- the kernel is measured against both
            nominal = cus * 4 * 2.4e9 / 4.0
            achievable = cus * 4 * 4.85e8
            rate = per_photon * value_per_gpu
            side.update(valu_issue_rate_per_s=rate, valu_issue_peak_per_s=nominal, valu_issue_frac=rate / nominal,
                        valu_issue_achievable_per_s=achievable, valu_issue_frac_of_achievable=rate / achievable)
            # VALU busy WHILE THE LAUNCHES OVERLAP: the instruction count per photon is a property of the work (PMC,
            # the same whether dispatches are serialised or not), the photon rate is measured here, and so is the shader
            # clock: every workgroup of the timed windows reads s_memtime against the constant 100 MHz s_memrealtime when
            # it starts and when it leaves (pvt_scene_clock) -- the clock the launches of THIS run ran at, overlapping.
            if clock and clock.get("shader_clock_mhz"):
                mhz = clock["shader_clock_mhz"]
                side["valu_busy_under_overlap"] = rate * 4.0 / (cus * 4 * mhz * 1e6)
                if pmc_per_photon and pmc_per_photon != per_photon:
                    side["valu_busy_under_overlap_upper"] = pmc_per_photon * value_per_gpu * 4.0 / (cus * 4 * mhz * 1e6)
                side["shader_clock_mhz_measured_in_this_run"] = mhz
                side["shader_clock_is"] = ("sum of s_memtime cycles / sum of s_memrealtime ticks (100 MHz) over the lives of "
                                           "all workgroups of the timed windows (pvt_scene_clock)")
        return summary.get("hbm_bytes_per_launch"), side
    except Exception:
        return None, ({"measured_in_this_run": False, "in_kernel": live} if live else None)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    # (ranks started by somebody else's launcher -- the driver's torch.distributed.run line -- get the same setting
    # spawn_ranks gives its own: dmabuf IPC, without which RCCL cannot share device memory across processes here)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch

    import __graft_entry__ as entry
    from benchmarks.configs import CONFIGS
    from pvtrace_amd.engine import compile_scene, native
    from pvtrace_amd.engine.compiler import EMIT_METHODS
    from pvtrace_amd.engine.emit import EmitterTables, emit_bundle
    from pvtrace_amd.engine.pipeline import BundlePipeline

    def die(msg):
        if rank == 0:
            print(json.dumps({"error": msg, "n_gpus": world}), flush=True)
        sys.exit(msg)

    if args.gpus != world:
        die(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks (python bench.py --gpus N does it itself)")
    cores_pinned = pin_rank_to_cores(local_rank, local_world)
    distributed = world > 1 or os.environ.get("PVT_BENCH_FORCE_DIST") == "1"  # (1-rank RCCL self-test)
    if not native.library_built():
        if world > 1:
            die("libpvtrace_hip.so is not built; run __graft_entry__.build() once before a multi-rank launch")
        entry.build()
    if not native.is_available():
        die("no MI355X visible: the engine has no CPU path (build ok, nothing to measure)")
    backend = os.environ.get("PVT_BENCH_BACKEND", "nccl")   # nccl = RCCL on ROCm
    n_devices = torch.cuda.device_count()
    if distributed and backend == "nccl" and local_world > n_devices:
        masks = {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if k in os.environ}
        die(f"{local_world} ranks on {n_devices} visible GPU(s): RCCL needs one GPU per rank"
            + (f" (device masks in the environment: {masks})" if masks else ""))
    local_rank %= n_devices   # (self-test: several ranks on a 1-GPU box, gloo backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = None
    if distributed:
        import torch.distributed as dist

        timeout = datetime.timedelta(seconds=args.rccl_timeout_s)
        # a rank that dies must surface on the others as an exception at their next collective (after the time-out), not
        # as a watchdog abort: rank 0 then still prints its line, with what it had measured and an `error` field
        os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev, timeout=timeout)
            else:
                dist.init_process_group(backend=backend, timeout=timeout)
            # a real collective before anything is timed: every rank contributes 1, the sum is what RCCL saw
            ones = torch.ones(1, dtype=torch.int64, device=dev)
            dist.all_reduce(ones)
            torch.cuda.synchronize(dev)
            rccl_ranks = int(ones.item())
        except Exception as exc:   # noqa: BLE001 -- say WHICH step failed, on the one line the driver reads
            die(f"process group ({backend}, {world} ranks) failed on rank {rank}: {type(exc).__name__}: {exc}")
        if rccl_ranks != world:
            die(f"all-reduce saw {rccl_ranks} ranks, expected {world}")

    n = args.photons
    nbuf = max(1, args.ray_buffers)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    tail_wide = int(os.environ.get("PVT_TAIL_WIDE", "1"))   # the last bundle of a window at full launch width

    def fence(pipe):
        pipe.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    rank_seconds = {}   # leg -> every rank's own seconds for its last window (one collective gives the MAX and the spread)

    def max_over_ranks(dt, leg=None):
        if distributed:
            t = torch.zeros(world, dtype=torch.float64, device=dev)
            t[rank] = dt
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            each = [float(v) for v in t.tolist()]
            if leg is not None:
                rank_seconds[leg] = each
            dt = max(each)
        return dt

    class Leg:
        """One config resident on this rank's GPU with its pipeline.  cfg2 is traced from rays resident in HBM
        (array-input mode, the reference's three arrays); the other configs sample their lights on the device."""

        def __init__(self, name, photons):
            spec = CONFIGS[name]
            self.name, self.spec, self.n = name, spec, photons
            self.scene = spec["build"]()
            self.compiled = compile_scene(self.scene)
            self.method = EMIT_METHODS[spec["emit_method"]]
            self.array_input = name == "cfg2" or os.environ.get("PVT_BENCH_ARRAY_INPUT") == "1"   # (developer A/B)
            self.ray_sets, self.host_rays = [None], None
            if self.array_input:
                # each rank's shard: global indices [rank*n, (rank+1)*n).  The steps rotate through `--ray-buffers`
                # distinct ray sets so that a step's input does not sit in the Infinity Cache from the previous step.
                # The sets are the same on every rank (a ray's history is decided by its RNG stream, seed + GLOBAL
                # index, which differs from rank to rank): the strong-scaling job below can then be the SAME job --
                # photon j starts as row j % n of set (j // n) % sets -- whatever the number of ranks that share it.
                self.ray_sets = []
                for b in range(nbuf):
                    p_, d_, w_, _ = emit_bundle(self.scene, photons, seed=1000 + 7919 * b)
                    if b == 0:
                        self.host_rays = (p_, d_, w_)
                    self.ray_sets.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (p_, d_, w_)))
                self.dscene = native.DeviceScene(self.compiled, device=local_rank)
            else:
                self.dscene = native.DeviceScene(self.compiled, device=local_rank,
                                                 emitter=EmitterTables(self.scene, strict=True))
            # Steps are independent bundles; like any streaming consumer of the engine they go through the
            # product's BundlePipeline (engine/pipeline.py): bundle k+1 is enqueued on another HIP stream while
            # bundle k drains, every bundle is fully traced and accumulated; the totals are summed over the ranks
            # before the closing fence.  --streams 1 gives the strictly serial schedule.
            self.pipe = BundlePipeline(self.dscene, depth=args.streams, distributed=distributed, reduce=args.reduce)
            self.pipe.wait_for_inputs()
            self.submit_s, self.submit_n = 0.0, 0   # host time spent enqueueing bundles (Python + the C ABI call)

        def step(self, k, timed, tail=False, m=None, offset=None, seed=None, closing=False):
            m = self.n if m is None else m
            rays = self.ray_sets[k % len(self.ray_sets)]
            if rays is not None and m != self.n:
                rays = tuple(t[:m] for t in rays)
            t_sub = time.perf_counter()
            self.pipe.submit(rays, m, seed=12345 + k * world * self.n if seed is None else seed,
                             ray_offset=rank * self.n if offset is None else offset, emit_seed=4242 + k * world * self.n,
                             maxsteps=1000, max_events=128, emit_method=self.method, timed=timed, tail=tail,
                             closing=closing)
            self.submit_s += time.perf_counter() - t_sub
            self.submit_n += 1

        def window(self, first_step, steps, timed_events=False):
            """One independent window of `steps` steps, fenced on both sides -> seconds (max over ranks)."""
            self.pipe.reset_totals()
            fence(self.pipe)
            t0 = time.perf_counter()
            for k in range(steps):   # (the last bundle on each stream finishes the photons carried along its stream)
                self.step(first_step + k, timed_events, tail=k >= steps - tail_wide, closing=k >= steps - args.streams)
            self.pipe.reduce_totals()   # fold the streams' totals; RCCL all-reduce over the ranks (reduce="end")
            fence(self.pipe)
            return max_over_ranks(time.perf_counter() - t0, leg=self.name)

        def spin_up(self, seconds, warmup):
            # Clocks: an idle MI355X needs tens of milliseconds of work to reach its sustained clocks, far more
            # than `--warmup` steps of ~0.5 ms provide; spin the same path up first (untimed, not counted).
            # First uses first: the first timed launch (HIP events with timestamps) and the first full-width
            # launch each stall the queue for tens of milliseconds once per process; paid here, BEFORE the clocks
            # are spun up, so that the --warmup steps run on a warm pipeline at sustained clocks like the timed ones.
            if seconds > 0:
                for k in range(max(2, tail_wide + 1)):
                    self.step(2_000_000 + k, True, tail=k >= 1)
                self.pipe.reduce_totals()
                self.pipe.synchronize()
                self.pipe.reset_totals()   # (totals reduced over the ranks are final: the pipeline refuses more bundles)
            t_spin, k_spin = time.perf_counter(), 0
            while time.perf_counter() - t_spin < seconds:
                for _ in range(20):
                    self.step(1_000_000 + k_spin, False)
                    k_spin += 1
                self.pipe.synchronize()
            for k in range(warmup):   # same path as the timed steps (events and the tail launch included)
                self.step(k, True, tail=k >= warmup - tail_wide)
            self.pipe.reduce_totals()   # also warms the RCCL communicator up (its first collective is slow)

        def fractions(self, photons):
            totals = self.pipe.totals_host()
            names = self.compiled.recorder_names
            return {names[i]: float(totals["rec_distinct"][i]) / photons for i in range(len(names))}

        def close(self):
            self.dscene.close()

    # ------------------------------------------------------------------ main loop (the contract's K steps)
    leg = Leg(args.config, n)
    leg.spin_up(args.spinup_s, args.warmup)
    leg.dscene.counters(reset=True)   # the kernel's own step counters, over the timed windows (read after the last one)
    first_dt = leg.window(args.warmup, args.steps, timed_events=True)
    kernel_ms = leg.pipe.kernel_ms()
    mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    fractions = leg.fractions(n * world * args.steps)
    launch = leg.dscene.launch_info()
    next_step = args.warmup + args.steps
    window_dts = [first_dt]
    for _ in range(max(0, args.repeats)):
        window_dts.append(leg.window(next_step, args.steps))
        next_step += args.steps
    in_kernel = live_counters(leg.dscene.counters(), n * args.steps * len(window_dts))   # (this rank's photons)
    main_clock = leg.dscene.clock()   # the shader clock of the timed windows' launches (reset with the counters above)
    ordered = sorted(window_dts)
    median_dt = ordered[(len(ordered) - 1) // 2]   # median (the slower of the two middle ones for an even count)
    per_window = n * world * args.steps
    value = per_window / median_dt
    # what a step costs the HOST of each rank (Python + ctypes + the enqueue): N ranks share the box's cores, and a step
    # of the stream is ~0.3 ms of GPU time, so this is the budget a multi-rank run lives on
    submit_us = [leg.submit_s / max(leg.submit_n, 1) * 1e6]
    if distributed:
        t_sub = torch.zeros(world, dtype=torch.float64, device=dev)
        t_sub[rank] = submit_us[0]
        dist.all_reduce(t_sub, op=dist.ReduceOp.SUM)
        submit_us = [float(v) for v in t_sub.tolist()]

    errors = []
    if os.environ.get("PVT_BENCH_DIE_AFTER_MAIN") == str(rank):   # (tests: a rank lost after the timed region)
        os._exit(3)

    def attempt(what, body):
        """The legs after the contract's K steps: a failure (a rank gone, a collective timed out) is recorded, not fatal."""
        try:
            return body()
        except Exception as exc:   # noqa: BLE001
            errors.append(f"{what}: {type(exc).__name__}: {exc}"[:400])
            return None

    state = {"next_step": next_step}

    def sustained_leg():
        # a short probe gives the step time without the fixed cost of a 20-step window
        probe_steps = 10 * args.steps
        probe = (leg.window(state["next_step"], probe_steps) if args.sustained_s >= 1.0
                 else median_dt / args.steps * probe_steps)
        state["next_step"] += probe_steps
        sus_steps = max(args.steps, int(args.sustained_s / (probe / probe_steps)))
        dt = leg.window(state["next_step"], sus_steps)
        state["next_step"] += sus_steps
        out = {"steps": sus_steps, "photons": n * world * sus_steps, "seconds": dt, "value": n * world * sus_steps / dt}
        if distributed and leg.name in rank_seconds:
            each = rank_seconds[leg.name]
            out["seconds_per_rank"] = each        # every rank's own barrier-to-barrier time for this leg
            out["rank_spread"] = (max(each) - min(each)) / max(each)
        return out

    def strong_leg(sustained):
        # BASELINE configs[2]: ONE job of `total` photons, rank r traces the index range
        # [r*total/world, (r+1)*total/world) in bundles of n (ray seed = seed + global index), the tallies
        # are all-reduced once at the end; time = barrier to barrier, max over ranks
        from pvtrace_amd.engine.distributed import shard_range

        def job(pipe, lo, hi, fenced):
            pipe.reset_totals()
            fenced()
            t0 = time.perf_counter()
            at = lo
            while at < hi:
                # bundles end on the job's own multiples of n, so that photon j is row j % n of ray set (j // n) % sets
                # for every way of sharding the job (a shard that starts in the middle of a bundle begins with a short one)
                g, off = divmod(at, n)
                m = min(n - off, hi - at)
                rays = leg.ray_sets[g % len(leg.ray_sets)]
                if rays is not None and (off or m != leg.n):
                    rays = tuple(t[off:off + m] for t in rays)
                pipe.submit(rays, m, seed=777, ray_offset=at, emit_seed=4242, maxsteps=1000, max_events=128,
                            emit_method=leg.method, timed=False, tail=at + m >= hi, closing=at + m * args.streams >= hi)
                at += m
            pipe.reduce_totals()
            fenced()
            return time.perf_counter() - t0

        lo, hi = shard_range(args.total_photons, rank, world)
        dt = max_over_ranks(job(leg.pipe, lo, hi, lambda: fence(leg.pipe)), leg="strong")
        st = leg.pipe.totals_host()
        names = list(leg.compiled.recorder_names)
        tallied = None
        if "entering" in names and "reflected" in names:
            tallied = int(st["rec_distinct"][names.index("entering")] + st["rec_distinct"][names.index("reflected")])
        strong = {"scaling": "strong", "total_photons": args.total_photons, "seconds": dt,
                  "value": args.total_photons / dt, "photons_tallied": tallied}
        job_ints = {k: np.asarray(st[k]).astype(np.int64) for k in ("rec_distinct", "rec_crossings", "rec_bins")}
        if distributed and "strong" in rank_seconds:
            strong["seconds_per_rank"] = rank_seconds["strong"]
        if sustained is not None:
            # What to expect at N GPUs: a rank traces total/(N n) bundles at the sustained step time, plus the part
            # of a job that does not shrink with N -- the ramp and the drain of its last bundles, the fold of the
            # streams' totals, the all-reduce and the fences -- measured here as what a --steps window costs beyond
            # its steps at the sustained rate.
            t_step = sustained["seconds"] / sustained["steps"]
            t_fixed = max(median_dt - args.steps * t_step, 0.0)
            bundles = args.total_photons / n
            strong["predicted"] = {
                "model": "T(N) = bundles/N * t_step + t_fixed; efficiency = T(1) / (N T(N))",
                "t_step_ms": t_step * 1e3, "t_fixed_ms": t_fixed * 1e3,
                "efficiency": {str(g): (bundles * t_step + t_fixed) / (g * (bundles / g * t_step + t_fixed))
                               for g in (2, 4, 8)},
            }
        if world > 1:
            # ... and what it IS at this N: rank 0 traces the whole job alone (its own pipeline, no collective) while
            # the others wait; efficiency = T(1) / (N T(N)), both measured in this run
            alone = None
            if rank == 0:
                solo = BundlePipeline(leg.dscene, depth=args.streams, distributed=False)
                solo.wait_for_inputs()
                for _ in range(2):   # (the first pass pays for the new pipeline's streams and buffers)
                    alone = job(solo, 0, args.total_photons, lambda: (solo.synchronize(), torch.cuda.synchronize(dev)))
                # the same job traced by ONE rank: its integer tallies must be the N-rank job's, bit for bit (the
                # reference's analogue: output independent of the thread count, tests/test_engine.py:169-176)
                one = solo.totals_host()
                strong["integer_tallies_equal_single_rank"] = bool(all(
                    np.array_equal(np.asarray(one[k]).astype(np.int64), job_ints[k]) for k in job_ints))
                strong["rec_distinct"] = [int(v) for v in job_ints["rec_distinct"]]
                solo.close()
            fence(leg.pipe)
            if rank == 0:
                strong["measured"] = {"seconds_one_gpu": alone, "seconds_n_gpus": dt, "n_gpus": world,
                                      "efficiency": alone / (world * dt)}
        return strong

    # ------------------------------------------------------------------ the other configs, same measurement
    def config_leg(name):
        bundles = max(1, args.config_photons // n)
        other = Leg(name, n)
        try:
            other.spin_up(min(args.spinup_s, 0.1), 2)
            dts, kms = [], []
            other.dscene.counters(reset=True)
            for w in range(5):
                dts.append(other.window(100 + w * bundles, bundles, timed_events=True))
                kms += other.pipe.kernel_ms()
            live = live_counters(other.dscene.counters(), n * bundles * 5)
            clock = other.dscene.clock()
            frac = other.fractions(n * world * bundles)
            dts.sort()
            photons = n * world * bundles
            v = photons / dts[len(dts) // 2]
            sus = None
            if args.config_sustained_s > 0:
                sus_steps = max(bundles, int(args.config_sustained_s / (dts[len(dts) // 2] / bundles)))
                dt = other.window(100_000, sus_steps)
                sus = {"steps": sus_steps, "photons": n * world * sus_steps, "seconds": dt, "value": n * world * sus_steps / dt}
            _, side = load_pmc(name, v / world, cus, live, clock)
            return {
                "workload": CONFIGS[name]["workload"], "photons_per_gpu": n * bundles, "bundles": bundles,
                "emission": "device (sampled by the wave that claims a chunk of rays, in the trace kernel)", "bundles_in_flight": args.streams,
                "value": v, "unit": "photons/s", "windows": len(dts), "min": photons / dts[-1], "max": photons / dts[0],
                "ms_per_window": dts[len(dts) // 2] * 1e3, "kernel_ms_mean": sum(kms) / len(kms),
                "launch": other.dscene.launch_info(), "instruction_side": side, "tallies": frac, "sustained": sus,
                "note": "a fenced window ends with the longest history of its last bundles traced alone (cfg4: photons "
                        "trapped by total internal reflection until the step limit, ~1 in 10^7: 1000 steps at ~3.8 us each, DESIGN.md §6); "
                        "`sustained` is the same stream without intermediate fences",
            }
        finally:
            other.close()

    # ------------------------------------------------------------------ triangle meshes (extension): photons/s against the face count
    def mesh_leg(sub):
        name = f"mesh{sub}"
        other = Leg(name, n)
        try:
            other.spin_up(min(args.spinup_s, 0.1), 2)
            other.dscene.counters(reset=True)
            dts = sorted(other.window(100 + w * 10, 10) for w in range(3))
            live = live_counters(other.dscene.counters(), n * 10 * 3)
            sus_steps = max(10, int(0.7 / (dts[1] / 10)))
            dt = other.window(100_000, sus_steps)
            return {"faces": 20 * 4 ** sub, "value": n * world * 10 / dts[1], "sustained": n * world * sus_steps / dt,
                    "unit": "photons/s", "launch": other.dscene.launch_info(), "in_kernel": live}
        finally:
            other.close()

    # ------------------------------------------------------------------ scene size: photons/s against the node count
    def size_leg(k):
        name = f"tiles{k}"
        other = Leg(name, n)
        try:
            other.spin_up(min(args.spinup_s, 0.1), 2)
            other.dscene.counters(reset=True)
            dts = sorted(other.window(100 + w * 10, 10) for w in range(3))
            live = live_counters(other.dscene.counters(), n * 10 * 3)
            clock = other.dscene.clock()
            v = n * world * 10 / dts[1]
            sus_steps = max(10, int(1.0 / (dts[1] / 10)))
            dt = other.window(100_000, sus_steps)
            _, side = load_pmc(name, v / world, cus, live, clock)
            return {
                "nodes": k * k + 1, "value": v, "sustained": n * world * sus_steps / dt, "unit": "photons/s",
                "launch": other.dscene.launch_info(), "node_grid": native.node_grid_plan(other.compiled) is not None,
                "valu_wave_instructions_per_photon": (side or {}).get("valu_wave_instructions_per_photon"),
                "valu_lane_utilisation": (side or {}).get("valu_lane_utilisation"),
                "in_kernel": live,
            }
        finally:
            other.close()

    # ------------------------------------------------------------------ the reference's one published engine figure
    def readme_leg():
        """`engine.simulate(scene, 1_000_000)` with DEFAULT arguments on hello_world -- the call behind the reference README's
        "460 k rays/s" (README.md:163-170; `record_every=1, max_events=128`: every event of every ray is kept).  Reported the
        reference's way (`elapsed` wraps the trace alone, api.py:232-245), end to end (emission, trace, the written rows of
        the event log brought to the host), and beside the CPU port on the same call (a bounded sample: 10^5 rays, whose dense
        log is 1.5 GB on the host; 10^6 would be 15 GB of mostly untouched rows)."""
        from benchmarks.configs import hello_world
        from pvtrace_amd import engine

        scene = hello_world()
        rays = 1_000_000
        engine.simulate(scene, 2000)   # the library, the scene's residency
        best, walls = None, []
        for rep in range(4):
            t0 = time.perf_counter()
            result = engine.simulate(scene, rays)
            wall = time.perf_counter() - t0
            walls.append(wall)
            events = int(result.data["counts"].sum())
            if best is None or wall < best["end_to_end_s"]:
                best = {"end_to_end_s": wall, "trace_s": result.elapsed, "kernel_ms": result.kernel_ms, "events": events}
            del result   # (its pinned blocks go back to torch's host allocator and serve the next call)
        out = {"call": "engine.simulate(hello_world, 1_000_000)  # defaults: record_every=1, max_events=128, emission auto (device)",
               "rays": rays, "events_kept": best["events"],
               "trace_only_rays_per_s": rays / best["trace_s"], "end_to_end_rays_per_s": rays / best["end_to_end_s"],
               "trace_ms": best["trace_s"] * 1e3, "end_to_end_ms": best["end_to_end_s"] * 1e3, "kernel_ms": best["kernel_ms"],
               "published_reference": {"value": 460_000, "unit": "rays/s", "where": "reference README.md:163-170 (its own hardware, trace only)"},
               "end_to_end_is": "emission on the device + trace + download of the written rows into pinned host blocks (packed; "
                                "the dense rows = rays x max_events columns of the reference are built on demand); best of 4 calls",
               "end_to_end_ms_each_call": [w * 1e3 for w in walls],
               "first_call_note": "the first call of a size also page-locks its result blocks (~0.5 GB here); later calls "
                                  "reuse the blocks the dropped result gave back"}
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            from pvtrace_amd.engine import compile_scene as compile_
            from pvtrace_amd.engine.emit import emit_bundle as emit_

            m = 100_000
            c = compile_(scene)
            p_, d_, w_, _ = emit_(scene, m, seed=3)
            cores = usable_cores()
            O.trace_bundle(c, p_[:2000], d_[:2000], w_[:2000], 1, 1000, 128, 0, cores, 1)
            t0 = time.perf_counter()
            O.trace_bundle(c, p_, d_, w_, 1, 1000, 128, 0, cores, 1)
            dt = time.perf_counter() - t0
            out["cpu_port"] = {"value": m / dt, "unit": "rays/s", "cores": cores, "kind": "port",
                               "sample": f"{m} rays of the same call (record_every=1, max_events=128: its dense log is allocated "
                                         f"and filled like the reference's, _kernel.pyx:1035-1047), {cores} OpenMP threads, {dt:.2f} s"}
        return out

    # ------------------------------------------------------------------ the C entry with HOST arrays (PCIe inside the call)
    def host_arrays_leg():
        """`_kernel.trace_bundle(compiled, positions, directions, wavelengths, ...)` on numpy arrays, as the reference's
        api.py:232-245 calls its kernel: `pvt_trace_bundle` uploads the rays (56 B/photon over PCIe, in chunks, a chunk
        traced while the next one is on its way), traces, brings the tallies back.  Wall time of the whole call."""
        from pvtrace_amd.engine import _kernel

        out = {"call": "pvtrace_amd.engine._kernel.trace_bundle(compiled, pos, dirs, wl, seed, 1000, 128, 0, 1, 0)  # numpy in, numpy out",
               "scene": CONFIGS["cfg2"]["workload"], "pcie_bytes_per_photon": 56, "sizes": {}}
        pos, dirs, wl = leg.host_rays
        _kernel.trace_bundle(leg.compiled, pos[:1000], dirs[:1000], wl[:1000], 1, 1000, 128, 0, 1, 0)
        for m in (1_000_000, 4_000_000):
            reps = -(-m // len(wl))
            p_, d_, w_ = (np.ascontiguousarray(np.concatenate([a] * reps)[:m]) for a in (pos, dirs, wl))
            walls = []
            for rep in range(7):
                t0 = time.perf_counter()
                _kernel.trace_bundle(leg.compiled, p_, d_, w_, 11 + rep, 1000, 128, 0, 1, 0)
                walls.append(time.perf_counter() - t0)
            walls.sort()
            out["sizes"][str(m)] = {"best_ms": walls[0] * 1e3, "median_ms": walls[len(walls) // 2] * 1e3,
                                    "photons_per_s": m / walls[0], "photons_per_s_median": m / walls[len(walls) // 2],
                                    "pcie_floor_ms_at_55GBs": 56 * m / 55e9 * 1e3}
        return out

    printed = threading.Lock()
    line_is_out = threading.Event()   # set once the line has been written (whoever wrote it)

    def report(more_errors=()):
        """Rank 0's ONE line, from whatever has been measured when it is called: at the end of the run, or -- from the
        watcher thread below -- when the launcher takes the job down because another rank is gone."""
        if not printed.acquire(blocking=False):
            return
        sustained, strong, extra, scaling = done["sustained"], done["strong"], done["extra"], done["scaling"]
        failures = list(errors) + list(more_errors)
        achieved = ALGORITHMIC_BYTES_PER_PHOTON * n / (mean_kernel_ms * 1e-3) / 1e9 if leg.array_input else 0.0
        traffic, instruction_side = load_pmc(args.config, value / world, cus, in_kernel, main_clock)
        rates = [per_window / d for d in window_dts]
        side = instruction_side or {}
        busy = side.get("valu_busy_under_overlap") or side.get("valu_issue_frac")
        hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "traffic": traffic,
               "achieved_at_step_rate": (ALGORITHMIC_BYTES_PER_PHOTON if leg.array_input else 0) * per_window / world / median_dt / 1e9,
               "note": "the figure the metric asks for: 56 algorithmic B/photon x photons per launch / mean launch duration; "
                       "not the binding resource"}
        out = {
            "metric": "photons/sec on 5x5x1 cm Lumogen-F-Red LSC",
            "value": value,
            "unit": "photons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": median_dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": CONFIGS[args.config]["workload"],
                "scene_builder": "pvtrace_amd.LSC((5,5,1)) + engine.instrument.face_recorders() (benchmarks/configs.py)"
                                 if args.config == "cfg2" else "benchmarks/configs.py",
                "photons_per_gpu_per_step": n,
                "sharding": (f"index-range x{world}, tallies {'RCCL' if backend == 'nccl' else backend} all-reduce "
                             + ("once per job, inside the timed region" if args.reduce == "end" else "per step"))
                            if distributed else "single GPU",
                "input": (f"rays resident in HBM (array-input mode, 56 B/photon), steps rotate through {nbuf} distinct "
                          f"ray sets ({nbuf * 56 * n / 1e6:.0f} MB)") if leg.array_input
                         else "device-side emission inside the trace kernel (0 B/photon in)",
                "bundles_in_flight": args.streams,
                "value_is": f"median of {len(window_dts)} fenced windows of {args.steps} steps",
            },
            "rccl_ranks": rccl_ranks,
            "host": {"usable_cores": usable_cores(), "cores_of_rank0": cores_pinned,
                     "submit_us_per_step_per_rank": submit_us, "gpu_us_per_step": median_dt / args.steps * 1e6},
            "roofline": {
                # The operative ceiling of this kernel is FP64 VALU issue (one wave64 instruction per SIMD every four
                # cycles); HBM, which the metric names, is kept beside it (`hbm`, same fields as before).
                "bound": "fp64-valu-issue" if busy else "hbm",
                "achieved": side.get("valu_issue_rate_per_s") if busy else achieved,
                "peak": side.get("valu_issue_peak_per_s") if busy else HBM_PEAK_GBS,
                "unit": "wave64 VALU instructions/s" if busy else "GB/s",
                "frac": busy if busy else achieved / HBM_PEAK_GBS,
                "frac_is": ("VALU busy with the launches overlapping: vector instructions per photon (PMC per trip x trips "
                            "counted in this run) x photons/s x 4 cycles / (SIMDs x shader clock read by the kernel in this run)")
                           if busy else "hbm",
                "traffic": traffic,
                "hbm": hbm,
                "steps_per_photon": in_kernel["steps_per_photon"] if in_kernel else None,
                # two different things: lanes HOLDING a live photon when a wave steps (counted by the kernel in this run),
                # and active lanes per vector instruction, divergence inside the step included (PMC pass)
                "live_lane_fraction": in_kernel["live_lane_fraction"] if in_kernel else None,
                "valu_lane_utilisation": side.get("valu_lane_utilisation"),
                "useful_fp64_lane_throughput_frac": (busy * side["valu_lane_utilisation"]) if busy and side.get("valu_lane_utilisation") else None,
                "kernel": "trace_kernel_w4<RECORD=0,TAB_LDS=1,SEENW=1,EMIT=%d> (pvt_trace_kernel.h trace_body, MESH=0)"
                          % (0 if leg.array_input else 1),
                "kernel_ms_mean": mean_kernel_ms,
                "instruction_side": instruction_side,
                "kernel_photons_per_s": n / (mean_kernel_ms * 1e-3),
                "note": "`value` is a STREAM of overlapping bundles; ONE launch of this size alone runs at kernel_photons_per_s "
                        "(its drain tail is latency-bound). kernel_ms_mean is per launch (HIP events on the launch's own "
                        "stream, first window); with several bundles in flight launches overlap, so ms_per_step < kernel_ms_mean",
            },
            "launch": launch,
            "tallies": fractions,
            "repeats": {"windows": len(rates), "steps_each": args.steps, "min": min(rates), "median": value,
                        "max": max(rates), "first_window": rates[0], "spread": (max(rates) - min(rates)) / value},
        }
        if sustained is not None:
            out["sustained"] = sustained
        if strong is not None:
            out["strong_scaling"] = strong
        if extra:
            out["configs"] = extra
        if scaling:
            out["scene_scaling"] = scaling
        if done["meshes"]:
            out["meshes"] = done["meshes"]
        if done.get("readme_case") or done.get("host_arrays"):
            out["extra"] = {k: done[k] for k in ("readme_case", "host_arrays") if done.get(k)}
        if failures:
            out["error"] = "; ".join(failures)   # (everything above was measured before the failure)
        if not args.no_cpu_baseline and world == 1 and leg.array_input:   # the CPU referee is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(leg.compiled, *leg.host_rays)
        print(json.dumps(out), flush=True)
        line_is_out.set()


    def watch_for_a_lost_rank():
        """torch.distributed.run ends the other ranks with SIGTERM as soon as one of them is gone -- possibly before rank 0
        has found out for itself (a collective that times out) and printed.  The signal is noted through a wake-up pipe,
        which works while the main thread sits inside a collective or a device synchronisation, and a thread prints the
        line with what there is."""
        if not (distributed and rank == 0):
            return
        r_fd, w_fd = os.pipe()
        os.set_blocking(w_fd, False)
        signal.signal(signal.SIGTERM, lambda signum, frame: None)   # (a Python-level handler: the byte is then written)
        signal.set_wakeup_fd(w_fd, warn_on_full_buffer=False)

        def waiter():
            while True:
                got = os.read(r_fd, 1)
                if got and got[0] == signal.SIGTERM:
                    report(["the launcher ended this rank (SIGTERM): another rank was lost during a leg after the timed region"])
                    # (the main thread may be in the middle of writing the line itself -- it reads the PMC summaries and hashes
                    # the library's code first, a few hundred milliseconds: leaving now would take the line with it)
                    line_is_out.wait(20.0)
                    os._exit(1)

        threading.Thread(target=waiter, daemon=True).start()

    # what the legs after the timed region have produced so far (rank 0's line is built from it, see report())
    done = {"sustained": None, "strong": None, "extra": {}, "scaling": None, "meshes": None, "readme_case": None, "host_arrays": None}
    watch_for_a_lost_rank()
    done["sustained"] = sustained = attempt("sustained leg", sustained_leg) if args.sustained_s > 0 and not errors else None
    done["strong"] = attempt("strong-scaling leg", lambda: strong_leg(sustained)) if args.total_photons > 0 and not errors else None
    extra = done["extra"]
    wanted = [c for c in args.extra_configs.split(",") if c and c != "none"] if args.config == "cfg2" else []
    for name in wanted:
        if errors:
            break
        got = attempt(f"config {name}", lambda: config_leg(name))
        if got is not None:
            extra[name] = got
    scaling = None
    sizes = ([int(k) for k in args.scene_sizes.split(",") if k and k != "none"]
             if args.config == "cfg2" and args.extra_configs != "none" else [])   # ('--extra-configs none': the main config only)
    if sizes and not errors:
        scaling = {"what": "k x k tile arrays of the headline slab in one world (benchmarks/configs.py: tiles_lsc), device "
                           "emission, same pipeline as the other configs; the reference intersects every node in every step "
                           "(_kernel.pyx:666-680), this engine walks a node grid from 8 nodes on (DESIGN.md)",
                   "photons_per_gpu_per_window": n * 10, "sizes": {}}
        done["scaling"] = scaling
        for k in sizes:
            if errors:
                break
            got = attempt(f"scene size tiles{k}", lambda: size_leg(k))
            if got is not None:
                scaling["sizes"][f"tiles{k}"] = got

    if sizes and not errors:   # (with the scene-size leg: '--extra-configs none' runs the main config only)
        done["meshes"] = meshes = {"what": "triangle-mesh extension (no reference counterpart: its engine rejects meshes): the glass "
                                           "ball of the reference's hello_world as an icosphere, device emission, same pipeline; "
                                           "stack-free BVH walk with the top of the tree in LDS (DESIGN.md §4.5)", "sizes": {}}
        for sub in (3, 5, 7):
            if errors:
                break
            got = attempt(f"mesh scene mesh{sub}", lambda: mesh_leg(sub))
            if got is not None:
                meshes["sizes"][f"mesh{sub}"] = got

    if world == 1 and args.config == "cfg2" and args.extra_configs != "none" and not errors:
        done["readme_case"] = attempt("readme case", readme_leg)
        if leg.array_input:
            done["host_arrays"] = attempt("host arrays", host_arrays_leg)

    if rank == 0:
        report()
    leg.close()
    if distributed and not errors:
        dist.barrier()
        dist.destroy_process_group()
    if errors:
        sys.stdout.flush()
        os._exit(1)   # (a lost rank leaves the process group unusable: no tidy shutdown to wait for)


if __name__ == "__main__":
    main()
