cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_train --output-format csv -- python3 tools/experiments/train_whole_time.py > gpurun_out/r06_prof_train.json 2> gpurun_out/r06_prof_train.err
python3 tools/trim_stats.py $(ls gpurun_out/r06_prof_train/*/*kernel_stats.csv | head -1) > gpurun_out/r06_train_step_rocprof_kernel_stats.txt
rm -rf gpurun_out/r06_prof_train
head -30 gpurun_out/r06_train_step_rocprof_kernel_stats.txt
