# Developer tool (GPU box): rocprofv3 kernel stats + PMC passes of the mesh probes (tools/gpu_perf.py mesh)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mesh_prof -o mesh -- python $R/tools/gpu_perf.py mesh > $R/gpurun_out/mesh_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/mesh_pmc_fetch -o pmc -- python $R/tools/gpu_perf.py mesh > $R/gpurun_out/mesh_pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/mesh_pmc_sq -o pmc -- python $R/tools/gpu_perf.py mesh > $R/gpurun_out/mesh_pmc_sq.log 2>&1
cd $R
grep -v amdgpu gpurun_out/mesh_prof.log | tail -8
head -4 gpurun_out/mesh_prof/mesh_kernel_stats.csv | cut -c1-200
