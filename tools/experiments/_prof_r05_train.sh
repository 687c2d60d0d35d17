cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_prof_train --output-format csv -- python3 tools/train_step_time.py > gpurun_out/r05_prof_train.log 2> gpurun_out/r05_prof_train.err
python3 tools/trim_stats.py $(ls gpurun_out/r05_prof_train/*/*kernel_stats.csv | head -1) > gpurun_out/r05_train_step_rocprof_kernel_stats.txt
rm -rf gpurun_out/r05_prof_train
cat gpurun_out/r05_prof_train.log; head -45 gpurun_out/r05_train_step_rocprof_kernel_stats.txt
python3 - <<'PY'
import re
tot=0; calls=0
for l in open('gpurun_out/r05_train_step_rocprof_kernel_stats.txt').read().splitlines()[1:]:
    m=re.match(r'(.{70})\s+(\d+)\s+([\d.]+)\s+([\d.]+)', l)
    if m: calls+=int(m.group(2)); tot+=float(m.group(3))
print("total kernel us", tot, "calls", calls)
PY
