#!/bin/bash
# Developer tool (GPU box): where a lone wave's step goes -- the -DPVT_STATS=1 build's per-section cycle counts on the
# scenes of tools/gpu_lone_step.py (every launch prints its line; the first of each group of six is a 2000-step launch)
export PVT_LIB=$PWD/build/dev/${1:-stats}.so
python tools/gpu_lone_step.py 2>&1 | grep -E "solo-wave|us per step|cfg4"
