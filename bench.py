"""Headline benchmark: photons/s on the 5x5x1 cm Lumogen-F-Red LSC (BASELINE.json
configs[1]), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
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

Rank 0 prints ONE JSON line; see the task contract for the fields.  `roofline` is
for the trace kernel: achieved = 56 algorithmic bytes/photon x photons per launch
/ mean launch duration (HIP events on the launch stream).  This path is NOT
HBM-bound: the fraction is reported because the metric asks
for it, next to instruction-side numbers that actually bound it.
`cpu_baseline` times the CPU referee (a port of the reference kernel, proven
bit-identical to it) on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
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


def main():
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
                    help="extra, independent timed windows of --steps steps after the headline one (spread estimate)")
    ap.add_argument("--sustained-s", type=float, default=1.2,
                    help="length of the sustained leg (back-to-back bundles) in seconds of GPU work; 0 = skip")
    ap.add_argument("--total-photons", type=int, default=100_000_000,
                    help="strong-scaling leg (BASELINE configs[2]): ONE job of this many photons split over the "
                         "ranks by index range, tallies all-reduced once; 0 = skip")
    ap.add_argument("--reduce", choices=("end", "bundle"), default="end",
                    help="multi-GPU: all-reduce the tallies once per job, inside the timed region "
                         "(default), or after every bundle")
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as entry
    from pvtrace_amd.engine import compile_scene, native
    from pvtrace_amd.engine.pipeline import BundlePipeline
    from pvtrace_amd.engine.emit import emit_bundle
    from tests import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    distributed = world > 1 or os.environ.get("PVT_BENCH_FORCE_DIST") == "1"  # (1-rank RCCL self-test)
    if not native.library_built():
        if world > 1:
            sys.exit("libpvtrace_hip.so is not built; run __graft_entry__.build() once before a multi-rank launch")
        entry.build()
    if not native.is_available():
        sys.exit("no MI355X visible: the engine has no CPU path (build ok, nothing to measure)")
    local_rank %= torch.cuda.device_count()   # (self-test: several ranks on a 1-GPU box, gloo backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        backend = os.environ.get("PVT_BENCH_BACKEND", "nccl")   # nccl = RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    n = args.photons
    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    # each rank's shard: global indices [rank*n, (rank+1)*n); emission seeded per shard and per buffer.
    # The steps rotate through `--ray-buffers` distinct ray sets so that a step's input does not sit in
    # the Infinity Cache from the previous step.
    nbuf = max(1, args.ray_buffers)
    ray_sets = []
    for b in range(nbuf):
        p_, d_, w_, _ = emit_bundle(scene, n, seed=1000 + rank + 7919 * b)
        if b == 0:
            pos, dirs, wl = p_, d_, w_
        ray_sets.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (p_, d_, w_)))
    dscene = native.DeviceScene(compiled, device=local_rank)
    # Steps are independent bundles; like any streaming consumer of the engine they go
    # through the product's BundlePipeline (engine/pipeline.py): bundle k+1 is enqueued on a
    # second HIP stream while bundle k drains, every bundle is fully traced and accumulated; the
    # totals are summed over the ranks before the closing fence.  --streams 1 gives the strictly serial schedule.
    pipe = BundlePipeline(dscene, depth=args.streams, distributed=distributed, reduce=args.reduce)
    pipe.wait_for_inputs()

    # the last bundle of a window is submitted as the job's tail (full launch width: +1 %, BundlePipeline.submit)
    tail_wide = int(os.environ.get("PVT_TAIL_WIDE", "1"))

    def step(k, timed, tail=False):
        pipe.submit(ray_sets[k % nbuf], n, seed=12345 + k * world * n, ray_offset=rank * n, maxsteps=1000,
                    max_events=128, emit_method=0, timed=timed, tail=tail)

    def fence():
        pipe.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Clocks: an idle MI355X needs tens of milliseconds of work to reach its sustained clocks, far more
    # than `--warmup` steps of ~0.5 ms provide; spin the same path up first (untimed, not counted).
    # First uses first: the first timed launch (HIP events with timestamps) and the first full-width launch each
    # stall the queue for tens of milliseconds once per process; paid here, BEFORE the clocks are spun up, so that
    # the --warmup steps below run on a warm pipeline at sustained clocks like the timed ones.
    if args.spinup_s > 0:
        for k in range(max(2, tail_wide + 1)):
            step(2_000_000 + k, True, tail=k >= 1)
        pipe.reduce_totals()
        pipe.synchronize()
    t_spin = time.perf_counter()
    k_spin = 0
    while time.perf_counter() - t_spin < args.spinup_s:
        for _ in range(20):
            step(1_000_000 + k_spin, False)
            k_spin += 1
        pipe.synchronize()
    for k in range(args.warmup):
        step(k, True, tail=k >= args.warmup - tail_wide)   # same path as the timed steps (events and the tail launch included); reset below
    pipe.reduce_totals()   # also warms the RCCL communicator up (its first collective is slow)
    pipe.reset_totals()
    fence()
    tic = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True, tail=k >= args.steps - tail_wide)
    pipe.reduce_totals()   # fold the streams' totals; RCCL all-reduce over the ranks (reduce="end")
    fence()
    elapsed = time.perf_counter() - tic
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = pipe.kernel_ms()
    mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    totals = pipe.totals_host()

    def window(first_step, steps, timed_events=False):
        """One more independent window of `steps` steps, fenced like the headline one -> seconds (max over ranks)."""
        pipe.reset_totals()
        fence()
        t0 = time.perf_counter()
        for k in range(steps):
            step(first_step + k, timed_events, tail=k >= steps - tail_wide)
        pipe.reduce_totals()
        fence()
        dt = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    next_step = args.warmup + args.steps
    repeat_values = []
    for r in range(max(0, args.repeats)):
        dt = window(next_step, args.steps)
        next_step += args.steps
        repeat_values.append(n * world * args.steps / dt)
    sustained = None
    if args.sustained_s > 0:
        est = elapsed / args.steps                      # seconds per step, from the headline window
        sus_steps = max(args.steps, int(args.sustained_s / est))
        dt = window(next_step, sus_steps)
        next_step += sus_steps
        sustained = {"steps": sus_steps, "photons": n * world * sus_steps, "seconds": dt,
                     "value": n * world * sus_steps / dt}
    strong = None
    if args.total_photons > 0:
        # BASELINE configs[2]: ONE job of `total` photons, rank r traces the index range
        # [r*total/world, (r+1)*total/world) in bundles of n (ray seed = seed + global index), the tallies
        # are all-reduced once at the end; time = barrier to barrier, max over ranks
        from pvtrace_amd.engine.distributed import shard_range

        lo, hi = shard_range(args.total_photons, rank, world)
        pipe.reset_totals()
        fence()
        t0 = time.perf_counter()
        at, k = lo, 0
        while at < hi:
            m = min(n, hi - at)
            pipe.submit(tuple(t[:m] for t in ray_sets[k % nbuf]), m, seed=777, ray_offset=at, maxsteps=1000,
                        max_events=128, emit_method=0, timed=False, tail=at + m >= hi)
            at += m
            k += 1
        pipe.reduce_totals()
        fence()
        dt = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        st = pipe.totals_host()
        strong = {"scaling": "strong", "total_photons": args.total_photons, "seconds": dt,
                  "value": args.total_photons / dt,
                  "photons_tallied": int(st["rec_distinct"][list(compiled.recorder_names).index("entering")]
                                         + st["rec_distinct"][list(compiled.recorder_names).index("reflected")])}

    if rank == 0:
        total_photons = n * world * args.steps
        value = total_photons / elapsed
        achieved = ALGORITHMIC_BYTES_PER_PHOTON * n / (mean_kernel_ms * 1e-3) / 1e9
        traffic, instruction_side = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):   # last committed PMC passes of this same command (tools/gpu_pmc.sh)
            try:
                summary = json.load(open(pmc))
                traffic = summary.get("hbm_bytes_per_launch")
                derived = summary.get("derived", {})
                instruction_side = {
                    "source": "profiles/pmc_summary.json (" + str(summary.get("stage", "")) + ")",
                    "valu_wave_instructions_per_photon": derived.get("valu_wave_instructions_per_photon"),
                    "valu_lane_utilisation": derived.get("valu_lane_utilisation"),
                    "wait_fraction_of_wave_cycles": derived.get("wait_any_fraction_of_wave_cycles"),
                    # measured with counters, for ONE launch of this shape alone (rocprofv3 serialises
                    # dispatches while sampling): 4*SQ_ACTIVE_INST_VALU / (SIMDs * GRBM_GUI_ACTIVE / 8)
                    "valu_busy_measured_single_launch": derived.get("valu_busy_measured"),
                }
                per_photon = derived.get("valu_wave_instructions_per_photon")
                if per_photon:
                    # the operative ceiling: VALU issue.  Nominal: one wave64 FP64 instruction per SIMD every 4
                    # cycles at 2.4 GHz.  ACHIEVABLE on this part with the kernel's four waves per SIMD: a pure
                    # chain of v_fma_f64 issues one per 4.83-5.26 nominal cycles (tools/gpu_fma_peak.hip,
                    # profiles/r02_fma_peak.txt) -- the kernel is measured against both
                    cus = torch.cuda.get_device_properties(dev).multi_processor_count
                    nominal = cus * 4 * 2.4e9 / 4.0
                    achievable = cus * 4 * 4.85e8          # wave-FMA/s/SIMD at 4 waves x 4 chains: 4.73e8 and 4.96e8 on two boxes
                    rate = per_photon * value / world
                    instruction_side["valu_issue_rate_per_s"] = rate
                    instruction_side["valu_issue_peak_per_s"] = nominal
                    instruction_side["valu_issue_frac"] = rate / nominal
                    instruction_side["valu_issue_achievable_per_s"] = achievable
                    instruction_side["valu_issue_frac_of_achievable"] = rate / achievable
            except Exception:
                traffic = None
        nrec = compiled.rec_node.shape[0]
        distinct = totals["rec_distinct"]
        names = compiled.recorder_names
        fractions = {names[i]: float(distinct[i]) / total_photons for i in range(nrec)}
        out = {
            "metric": "photons/sec on 5x5x1 cm Lumogen-F-Red LSC",
            "value": value,
            "unit": "photons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: 5x5x1 cm LSC-equivalent scene, Lumogen F Red 305 "
                            "(10 cm^-1 peak, qy 1) + 0.1 cm^-1 background, 20-degree cone @555 nm, "
                            "10 recorders, record_every=0, emit_method=kT, maxsteps=1000",
                "photons_per_gpu_per_step": n,
                "sharding": (f"index-range x{world}, tallies {'RCCL' if os.environ.get('PVT_BENCH_BACKEND', 'nccl') == 'nccl' else os.environ['PVT_BENCH_BACKEND']} all-reduce "
                             + ("once per job, inside the timed region" if args.reduce == "end" else "per step")) if distributed
                            else "single GPU",
                "input": f"rays resident in HBM (array-input mode, 56 B/photon), steps rotate through {nbuf} distinct "
                         f"ray sets ({nbuf * 56 * n / 1e6:.0f} MB)",
                "bundles_in_flight": args.streams,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "trace_kernel_w4<RECORD=0,TAB_LDS=1,SEENW=1,EMIT=0> (pvt_trace_kernel.h trace_body, MESH=0)",
                "kernel_ms_mean": mean_kernel_ms,
                "instruction_side": instruction_side,
                "kernel_photons_per_s": n / (mean_kernel_ms * 1e-3),
                "achieved_at_step_rate": ALGORITHMIC_BYTES_PER_PHOTON * n * args.steps / elapsed / 1e9,
                "note": "not HBM-bound: 56 algorithmic B/photon; the loop is FP64-VALU/latency/"
                        "divergence-bound (DESIGN.md). kernel_ms_mean is per launch (HIP events on the "
                        "launch's own stream); with several bundles in flight launches overlap, so "
                        "ms_per_step < kernel_ms_mean",
            },
            "launch": dscene.launch_info(),
            "tallies": fractions,
        }
        if repeat_values:
            rv = sorted(repeat_values + [value])
            out["repeats"] = {"windows": len(rv), "steps_each": args.steps, "min": rv[0], "median": rv[len(rv) // 2],
                              "max": rv[-1], "spread": (rv[-1] - rv[0]) / rv[len(rv) // 2]}
        if sustained is not None:
            out["sustained"] = sustained
        if strong is not None:
            out["strong_scaling"] = strong
        if not args.no_cpu_baseline and world == 1:   # the CPU referee is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(compiled, pos, dirs, wl)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
