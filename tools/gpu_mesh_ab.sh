#!/bin/bash
# Developer tool (GPU box): mesh dev builds (DEVV=2 tools/dev_build.sh <name>) side by side -- the carried stream of
# tools/gpu_mesh_stream.py, then the bit-for-bit mesh tests against the brute-force oracle with each build.
# usage: tools/gpu_mesh_ab.sh "mbase mq4" [rounds] [scenes...]
libs=$1; rounds=${2:-2}; shift; shift
scenes=${@:-mesh_lsc mesh_gem l_prism ico3 ico5 ico7}
export PVT_STREAM_STEPS=${PVT_STREAM_STEPS:-60}
for r in $(seq $rounds); do
  for lib in $libs; do
    echo "== $lib"
    PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$lib.so python tools/gpu_mesh_stream.py $scenes 2>&1 | grep photons
  done
done
for lib in $libs; do
  echo "== parity $lib"
  PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$lib.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "mesh or prism or gem or ico" 2>&1 | tail -2
done
