#!/bin/bash
# round 6, session 2: the whole GPU suite with its durations, then the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --durations=30 > gpurun_out/r06_pytest_gpu_c.log 2>&1
echo "pytest rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r06_pytest_gpu_c.log
tail -45 gpurun_out/r06_pytest_gpu_c.log
t0=$(date +%s)
timeout 600 python bench.py > gpurun_out/r06_bench_c.json 2> gpurun_out/r06_bench_c.err
echo "bench rc $? wall $(( $(date +%s) - t0 )) s"
tail -3 gpurun_out/r06_bench_c.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_c.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "stale", (d["roofline"].get("instruction_side") or {}).get("stale"))
print("configs", {k: (v.get("value"), v.get("sustained")) for k, v in d.get("configs", {}).items()})
print("readme", d.get("extra"))
print("cpu", d.get("cpu_baseline"))
PY
