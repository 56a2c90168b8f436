#!/bin/bash
# Developer tool (GPU box): shader clock and power while the bench's sustained leg runs
(timeout 60 python bench.py --no-cpu-baseline --repeats 0 --total-photons 0 --sustained-s 8 > /dev/null 2>&1) &
sleep 4
for i in 1 2 3 4 5 6; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics" | tr -s " " | head -3; sleep 1; done
wait
