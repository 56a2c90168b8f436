"""The N>1 device path with two real processes on the one GPU of the test box: each rank builds
its own scene on cuda:0, traces its index-range shard with the HIP kernel, and the tallies are
summed with a gloo all-reduce of the device tensors (RCCL refuses two ranks on one GPU, so the
collective backend is the only thing that differs from an 8-GPU run).  Both ranks must end up
with the single-process result, exactly for the integer tallies."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from pvtrace_amd.engine import BundlePipeline, compile_scene, native
    from pvtrace_amd.engine.distributed import shard_range, simulate_sharded
    from pvtrace_amd.engine.emit import EmitterTables
    from tests import scenes

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    scene = scenes.lsc_equivalent()
    sharded = simulate_sharded(scene, n, seed=8, emit_seed=2, record_every=70, max_events=64, device=0)
    assert sharded.shard == shard_range(n, rank, world, align=70) and sharded.shard[0] % 70 == 0
    out = {f"sharded_{k}": sharded.data[k] for k in ("rec_distinct", "rec_crossings", "rec_bins", "rec_sums", "counts")}

    # pipelined job: this rank's shard as 4 bundles, tallies reduced once at the end / per bundle
    start, stop = shard_range(n, rank, world)
    dscene = native.DeviceScene(compile_scene(scene), device=0, emitter=EmitterTables(scene))
    try:
        for mode in ("end", "bundle"):
            pipe = BundlePipeline(dscene, depth=3, distributed=True, reduce=mode)
            edges = np.linspace(start, stop, 5).astype(int)
            for a, b in zip(edges[:-1], edges[1:]):
                pipe.submit(None, int(b - a), seed=8, ray_offset=int(a), emit_seed=2, emit_method=0)
            totals = pipe.totals_host()
            for k in ("rec_distinct", "rec_crossings", "rec_bins", "rec_sums"):
                out[f"pipe_{mode}_{k}"] = totals[k]
    finally:
        dscene.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1800)
def test_two_processes_on_one_gpu_reproduce_the_single_process_job(tmp_path):
    import torch.multiprocessing as mp

    from pvtrace_amd import engine
    from tests import scenes

    n, world = 200_033, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    whole = engine.simulate(scenes.lsc_equivalent(), n, seed=8, record_every=70, max_events=64,
                            emission="device", emit_seed=2)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in ranks:
        for prefix in ("sharded", "pipe_end", "pipe_bundle"):
            for k in ("rec_distinct", "rec_crossings", "rec_bins"):
                assert np.array_equal(r[f"{prefix}_{k}"], whole.data[k]), (prefix, k)
            assert np.allclose(r[f"{prefix}_rec_sums"], whole.data["rec_sums"], rtol=1e-12)
    # sampled event logs are per shard; shards start on multiples of record_every (here 99 960,
    # not the even split 100 016), so together they are the single-process log
    counts = np.concatenate([r["sharded_counts"] for r in ranks])
    assert np.array_equal(counts, whole.data["counts"])


@pytest.mark.timeout(1800)
def test_bench_script_runs_with_two_ranks_and_reports_every_leg(tmp_path):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per process),
    the two ranks sharing the one GPU of the test box and reducing over gloo: weak-scaling line, repeat
    windows, sustained leg and the strong-scaling leg, whose reduced tallies must account for every photon."""
    import json
    import subprocess

    env = dict(os.environ, PVT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--photons", "20000", "--repeats", "2",
           "--sustained-s", "0.02", "--total-photons", "100001", "--ray-buffers", "2", "--spinup-s", "0",
           "--no-cpu-baseline", "--config-photons", "60000"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert done.returncode == 0, done.stderr[-2000:]
    line = [l for l in done.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    _check_two_rank_line(out)


def _check_two_rank_line(out):
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["photons_per_gpu_per_step"] == 20000 and "index-range x2" in out["config"]["sharding"]
    assert out["repeats"]["windows"] == 3 and out["sustained"]["photons"] >= 2 * 3 * 20000
    assert out["strong_scaling"]["total_photons"] == 100001 == out["strong_scaling"]["photons_tallied"]
    # the weak line's tallies are the sum over both ranks' steps: fractions of 2 * 3 * 20000 photons
    assert abs(out["tallies"]["entering"] + out["tallies"]["reflected"] - 1.0) < 1e-12
    assert out["rccl_ranks"] == 2                      # counted by a real all-reduce of ones
    assert out["value"] == out["repeats"]["median"] and out["repeats"]["min"] <= out["value"] <= out["repeats"]["max"]
    eff = out["strong_scaling"]["predicted"]["efficiency"]
    assert 0.0 < eff["8"] <= eff["4"] <= eff["2"] <= 1.0
    measured = out["strong_scaling"]["measured"]          # rank 0 alone against both ranks, same run
    assert measured["n_gpus"] == 2 and measured["seconds_one_gpu"] > 0 and 0.0 < measured["efficiency"] < 10.0   # (two ranks on ONE GPU: not a speed-up)
    assert len(out["strong_scaling"]["seconds_per_rank"]) == 2 and len(out["sustained"]["seconds_per_rank"]) == 2
    assert "error" not in out
    sizes = out["scene_scaling"]["sizes"]
    assert sizes["tiles11"]["nodes"] == 122 and sizes["tiles11"]["node_grid"] and not sizes["tiles1"]["node_grid"]
    for name, per_rank in (("cfg4", 60000), ("cfg5", 60000)):   # the other configs: 3 bundles of 20 000 per rank
        leg = out["configs"][name]
        assert leg["value"] > 0 and leg["photons_per_gpu"] == per_rank and leg["kernel_ms_mean"] > 0
    assert abs(out["configs"]["cfg4"]["tallies"]["exit"] - 1.0) < 1e-3      # nothing absorbs in nested_cylinders
    assert abs(out["configs"]["cfg5"]["tallies"]["top-reflect-map"] - 0.28) < 0.02   # mirror quadrant + 4 % Fresnel


@pytest.mark.timeout(1800)
def test_bench_script_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the script re-executes itself under
    torch.distributed.run and prints the same line.  This run keeps the clock spin-up of the
    default flags: it reduces the totals over the ranks before the timed part, which must not wedge the pipeline."""
    import json
    import subprocess

    env = dict(os.environ, PVT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--photons", "20000", "--repeats", "2", "--sustained-s", "0.02", "--total-photons", "100001",
           "--ray-buffers", "2", "--spinup-s", "0.05", "--no-cpu-baseline", "--config-photons", "60000"]   # (with the clock spin-up)
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert done.returncode == 0, done.stderr[-2000:]
    _check_two_rank_line(json.loads([l for l in done.stdout.splitlines() if l.startswith("{")][-1]))


def test_bench_refuses_two_rccl_ranks_on_one_gpu_with_a_readable_line():
    """RCCL needs one GPU per rank: on the 1-GPU test box `--gpus 2` with the nccl backend must end quickly with
    an error LINE on rank 0 (what the driver would read), not hang in communicator setup."""
    import json
    import subprocess

    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a single-GPU box")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PVT_BENCH_BACKEND", None)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                          cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert done.returncode != 0
    lines = [l for l in done.stdout.splitlines() if l.startswith("{")]
    assert lines and "one GPU per rank" in json.loads(lines[-1])["error"]


@pytest.mark.timeout(1800)
def test_bench_script_over_rccl_with_one_rank():
    """The collectives of an N-GPU run on the real backend (RCCL), as far as a one-GPU box can take them: the
    bench with its process group forced on for a single rank -- communicator with `device_id`, the counting
    all-reduce, the spin-up's reduction, barriers, the MAX over ranks, the tallies' all-reduce of every window."""
    import json
    import subprocess

    env = dict(os.environ, PVT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1")
    env.pop("PVT_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--photons", "50000", "--repeats", "2", "--sustained-s", "1.0", "--total-photons", "400001",
           "--ray-buffers", "2", "--spinup-s", "0.05", "--no-cpu-baseline", "--config-photons", "100000",
           "--config-sustained-s", "0.2"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert done.returncode == 0, done.stderr[-2000:]
    out = json.loads([l for l in done.stdout.splitlines() if l.startswith("{")][-1])
    assert out["rccl_ranks"] == 1 and out["n_gpus"] == 1 and out["value"] > 0
    assert out["strong_scaling"]["photons_tallied"] == 400001
    assert abs(out["tallies"]["entering"] + out["tallies"]["reflected"] - 1.0) < 1e-12
    assert out["configs"]["cfg4"]["sustained"]["value"] > 0 and out["configs"]["cfg5"]["value"] > 0


@pytest.mark.timeout(1800)
def test_rank_zero_still_prints_its_line_when_another_rank_is_lost_after_the_timed_region():
    """the driver reads ONE line from rank 0.  A rank that dies after the contract's K steps (here:
    made to, right after the main windows) must not take that line with it: rank 0 reports what it had measured,
    with an `error` field naming the leg that failed."""
    import json
    import subprocess

    env = dict(os.environ, PVT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PVT_BENCH_DIE_AFTER_MAIN="1")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--photons", "20000", "--repeats", "2", "--sustained-s", "0.02", "--total-photons", "100001",
           "--ray-buffers", "2", "--spinup-s", "0", "--no-cpu-baseline", "--config-photons", "60000", "--rccl-timeout-s", "30"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    lines = [l for l in done.stdout.splitlines() if l.startswith("{")]
    assert lines, done.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["rccl_ranks"] == 2      # the timed region is all there
    assert "error" in out and "leg" in out["error"]
    assert done.returncode != 0
