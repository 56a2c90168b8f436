#!/bin/bash
# Developer tool: build ablation variants (results WRONG by design; timing only) and time them on the GPU box.
cd /root/repo/pvtrace_amd/csrc
for v in 1 4 16 32 64; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -fPIC -shared -DPVT_ABLATE=$v pvt_trace.hip -o /root/repo/build/abl/abl_$v.so & done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -fPIC -shared -DPVT_STATS=1 pvt_trace.hip -o /root/repo/build/abl/stats.so &
wait
cd /root/repo; cp pvtrace_amd/csrc/libpvtrace_hip.so build/abl/abl_0.so
/usr/local/graft/bin/gpurun --timeout 900 -- 'for v in 0 1 4 16 32 64; do echo "== ablate $v"; PVT_LIB=$PWD/build/abl/abl_$v.so python tools/gpu_perf.py lsc 2>&1 | grep -v amdgpu.ids; done; PVT_LIB=$PWD/build/abl/stats.so python tools/gpu_perf.py lsc 2>&1 | grep "pvt stats" | sed -n "1p;7p"'
