cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export P=8 CPU=0 GRAPH=0
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_cfg5 --output-format csv -- python3 tools/bench_2d3d.py > gpurun_out/r06_prof_cfg5.json 2> gpurun_out/r06_prof_cfg5.err
python3 tools/trim_stats.py $(ls gpurun_out/r06_prof_cfg5/*/*kernel_stats.csv | head -1) > gpurun_out/r06_cfg5_p8_rocprof_kernel_stats_v2.txt
rm -rf gpurun_out/r06_prof_cfg5
head -12 gpurun_out/r06_cfg5_p8_rocprof_kernel_stats_v2.txt
