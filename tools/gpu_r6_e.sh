#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host_bundle.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/gpu_host_bundle.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_host_bundle.txt | grep default
timeout 600 python bench.py > gpurun_out/r06_bench_e.json 2> gpurun_out/r06_bench_e.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_e.json").read().strip().splitlines()[-1])
print("value", d["value"], "error", d.get("error"))
print(json.dumps(d.get("extra"), indent=1))
PY
