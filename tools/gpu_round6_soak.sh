#!/bin/bash
# Round 6, long differential run on the final tree (device code e9dccaf44e6fe39b): fuzz GPU vs oracle, many-node scenes
# through the node grid, the chunked host-buffer upload, carried photons.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "tools/gpu_fuzz.py 1300000 40000:"; timeout 2700 python tools/gpu_fuzz.py 1300000 40000 2>&1 | grep -v amdgpu | tail -2
echo "tools/gpu_fuzz_many.py 50000 1500:"; timeout 1500 python tools/gpu_fuzz_many.py 50000 1500 2>&1 | grep -v amdgpu | tail -2
echo "PVT_HOST_CHUNK_RAYS=300 tools/gpu_fuzz.py 1500000 6000:"; PVT_HOST_CHUNK_RAYS=300 timeout 1200 python tools/gpu_fuzz.py 1500000 6000 2>&1 | grep -v amdgpu | tail -2
echo "tools/gpu_soak_carry.py 600:"; timeout 800 python tools/gpu_soak_carry.py 600 2>&1 | grep -v amdgpu | tail -2
} 2>&1 | tee gpurun_out/r06_soak_long.txt
