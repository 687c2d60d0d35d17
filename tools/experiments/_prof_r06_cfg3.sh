cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export P=8 CPU=0 GRAPH=0
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_cfg3 --output-format csv -- python3 tools/bench_cfg3.py > gpurun_out/r06_prof_cfg3.json 2> gpurun_out/r06_prof_cfg3.err
python3 tools/trim_stats.py $(ls gpurun_out/r06_prof_cfg3/*/*kernel_stats.csv | head -1) > gpurun_out/r06_cfg3_rocprof_kernel_stats.txt
python3 tools/trace_by_grid.py gpurun_out/r06_prof_cfg3 pgemm > gpurun_out/r06_cfg3_pgemm_by_grid.txt
rm -rf gpurun_out/r06_prof_cfg3
head -16 gpurun_out/r06_cfg3_rocprof_kernel_stats.txt; cat gpurun_out/r06_cfg3_pgemm_by_grid.txt
