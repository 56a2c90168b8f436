#!/bin/bash
# Developer tool: register / spill metadata and SGPR-spill traffic (v_readlane/v_writelane) of the dev variants.
# usage: tools/isa_meta.sh [extra hipcc flags]   (writes /tmp/isa_meta.s)
cd /root/repo/pvtrace_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -mllvm -disable-machine-licm -fno-unroll-loops -DPVT_DEV_VARIANTS=1 "$@" \
    --cuda-device-only -S pvt_trace.hip -o /tmp/isa_meta.s 2>/dev/null
python3 - <<'PY'
import re
s = open('/tmp/isa_meta.s').read()
for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
    name = f.split("\n", 1)[0].strip()
    if "trace_kernel" not in name: continue
    body = f.split(".end_amdhsa_kernel")[0]
    v = len(re.findall(r"^\s+v_", body, re.M)); rl = body.count("v_readlane_b32"); wl = body.count("v_writelane_b32")
    sc = len(re.findall(r"^\s+s_", body, re.M))
    print(f"{name[:60]:60s} valu {v} (readlane {rl} writelane {wl}) salu {sc}")
for m in re.finditer(r"\.name:\s+(\S*trace_kernel\S*)(.*?)\.wavefront_size", s, re.S):
    meta = dict(re.findall(r"\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|agpr_count):\s+(\d+)", m.group(2)))
    print(m.group(1)[:60], meta)
PY
