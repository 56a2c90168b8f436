"""The multi-rank rehearsal of the bench on the tree as it is: `bench.py --gpus 8` as the driver
launches it -- torch.distributed.run, one process per rank -- with the eight ranks sharing the one GPU of the test box
and reducing over gloo (RCCL refuses two ranks on one GPU; the backend is the only thing that differs from an 8-GPU
node).  Every rank runs the product's pipeline -- carried photons, closing launches, the all-reduce inside the timed
region -- and the strong-scaling leg traces BASELINE configs[2], one 10^8-photon job sharded by index range: all of
its photons must be in the reduced tallies, and its integer tallies must be those of the same job traced by ONE
rank (the reference's analogue: output independent of the thread count, tests/test_engine.py:169-176; merge
_kernel.pyx:1099-1102; seed rule api.py:249-264).  No scaling curve is asked of a one-GPU box."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
def test_bench_with_eight_ranks_on_one_gpu(tmp_path):
    env = dict(os.environ, PVT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "8", "--steps", "4", "--warmup", "1", "--repeats", "1", "--sustained-s", "0",
           "--extra-configs", "none", "--scene-sizes", "none", "--no-cpu-baseline"]
    t0 = time.perf_counter()
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    wall = time.perf_counter() - t0
    assert done.returncode == 0, done.stderr[-3000:]
    out = json.loads([l for l in done.stdout.splitlines() if l.startswith("{")][-1])
    assert "error" not in out, out.get("error")
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8                       # counted by a real all-reduce of ones
    assert out["config"]["photons_per_gpu_per_step"] == 1_000_000 and "index-range x8" in out["config"]["sharding"]
    assert abs(out["tallies"]["entering"] + out["tallies"]["reflected"] - 1.0) < 1e-12   # 8 x 4 x 10^6 photons, all there
    strong = out["strong_scaling"]
    assert strong["total_photons"] == 100_000_000 == strong["photons_tallied"]
    assert strong["integer_tallies_equal_single_rank"] is True
    assert len(strong["seconds_per_rank"]) == 8 and strong["measured"]["n_gpus"] == 8
    # the host side of a step, per rank (eight Python submit loops on the box's cores against ~0.3 ms of GPU time a step)
    costs = out["host"]["submit_us_per_step_per_rank"]
    assert len(costs) == 8 and all(0.0 < c < 5000.0 for c in costs), costs
    # the kernel's own counters of rank 0's timed windows: the headline scene's loop count per photon
    assert abs(out["roofline"]["steps_per_photon"] - 6.92) < 0.07
    assert wall < 120.0, wall
    record = os.environ.get("PVT_EIGHT_RANKS_RECORD")   # (tools/gpu_round5.sh keeps the line under profiles/)
    if record:
        out["wall_s_of_the_whole_run"] = wall
        with open(record, "w") as fp:
            json.dump(out, fp)
