# Developer tool (GPU box): full GPU test suite, then the kernel-time probes
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15
python tools/gpu_perf.py lsc 2>&1 | grep -v amdgpu.ids
python tools/gpu_perf.py all 2>&1 | grep -v amdgpu.ids
