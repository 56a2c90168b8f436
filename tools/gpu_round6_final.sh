#!/bin/bash
# The closing pass of round 6 on the final tree: the GPU suite with its slowest tests, the bench with the driver's call (PMC
# summaries of this tree in place: `stale` must read false), the fuzz with the host-buffer entry forced to upload in chunks.
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/pytest_gpu_final.log 2>&1
echo "pytest -m gpu: rc $? wall $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/pytest_gpu_final.log
grep -E "passed|failed" gpurun_out/pytest_gpu_final.log | tail -3
timeout 900 python bench.py 2>gpurun_out/bench_final.err > gpurun_out/bench_final.json; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
s = d["roofline"]["instruction_side"]
print("value", d["value"], "frac", d["roofline"]["frac"], "stale", s.get("stale"), "clock", s.get("shader_clock_mhz_measured_in_this_run"), "error", d.get("error"))
PY
PVT_HOST_CHUNK_RAYS=400 timeout 1200 python tools/gpu_fuzz.py 900000 3000 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/fuzz_chunked.txt
