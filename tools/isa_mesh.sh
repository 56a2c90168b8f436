#!/bin/bash
# Developer tool: register / spill metadata of the MESH dev variants (-DPVT_DEV_VARIANTS=2).  usage: tools/isa_mesh.sh [extra flags]
cd /root/repo/pvtrace_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -mllvm -disable-machine-licm -fno-unroll-loops -DPVT_DEV_VARIANTS=2 "$@" \
    --cuda-device-only -S pvt_trace.hip -o /tmp/isa_mesh.s 2>/dev/null
python3 - <<'PY'
import re
s = open('/tmp/isa_mesh.s').read()
for m in re.finditer(r"\.name:\s+(\S*trace_kernel\S*)(.*?)\.wavefront_size", s, re.S):
    meta = dict(re.findall(r"\.(sgpr_spill_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", m.group(2)))
    print(m.group(1)[20:62], meta)
PY
