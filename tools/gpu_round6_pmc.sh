#!/bin/bash
# PMC passes of round 6: every config bench.py reads a summary for (each counter set in its own rocprofv3 run, --kernel-trace
# only: tools/gpu_pmc.sh), summarised with the built library's .text hash (tools/pmc_summarise.py).
for cfg in cfg2 cfg4 cfg5 tiles3 tiles6 tiles11; do
  tools/gpu_pmc.sh pipelined $cfg > gpurun_out/pmc_$cfg.log 2>&1
  tail -3 gpurun_out/pmc_$cfg.log
done
ls gpurun_out/*pmc_summary*.json
