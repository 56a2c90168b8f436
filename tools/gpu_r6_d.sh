#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host_bundle.py tests/test_gpu_engine_api.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15
timeout 300 python tools/gpu_host_bundle.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_host_bundle.txt
timeout 300 python tools/gpu_readme_case.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_readme_case.txt
