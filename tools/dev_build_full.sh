#!/bin/bash
# Developer tool: the FULL library (every variant, as __graft_entry__.build() compiles it) with extra flags, into
# build/dev/<name>.so -- for A/B runs that need device emission, grids or meshes.   usage: tools/dev_build_full.sh name [flags]
name=$1; shift
mkdir -p /root/repo/build/dev
cd /root/repo
python - "$name" "$@" <<'PY'
import subprocess, sys, os
import __graft_entry__ as g
out = os.path.join(g.ROOT, "build", "dev", sys.argv[1] + ".so")
subprocess.check_call(["/opt/rocm/bin/hipcc", *g.HIPCC_FLAGS, *sys.argv[2:], os.path.join(g.CSRC, "pvt_trace.hip"), "-o", out])
print(out)
PY
