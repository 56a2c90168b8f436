#!/bin/bash
for v in rec4 rec3; do echo "== $v"; PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$v.so python tools/gpu_history.py 2>/dev/null; PVT_LIB=$GRAFT_REPO_ROOT/build/dev/$v.so python tools/gpu_history_scaling.py 2>/dev/null | grep "n=1000000"; done
