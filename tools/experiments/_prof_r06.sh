# How the round-6 files under profiles/ were produced (gpurun calls; outputs land in gpurun_out/ and are copied into profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export P=8 CPU=0
STEP=${1:-all}
if [ $STEP = all ] || [ $STEP = stats ]; then
# 1. kernel stats of the default bench's loop                  -> r06_bench_default_rocprof_kernel_stats.{txt,csv}
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-other-configs > gpurun_out/r06_prof_bench.json 2> gpurun_out/r06_prof_bench.err
cp $(ls gpurun_out/r06_prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/r06_bench_default_rocprof_kernel_stats.csv
python3 tools/trim_stats.py gpurun_out/r06_bench_default_rocprof_kernel_stats.csv > gpurun_out/r06_bench_default_rocprof_kernel_stats.txt
rm -rf gpurun_out/r06_prof_bench
# 2. cfg5 (8 pairs per call) kernel stats on the final tree     -> r06_cfg5_p8_rocprof_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_prof_cfg5 --output-format csv -- python3 tools/bench_2d3d.py > gpurun_out/r06_prof_cfg5.json 2> gpurun_out/r06_prof_cfg5.err
python3 tools/trim_stats.py $(ls gpurun_out/r06_prof_cfg5/*/*kernel_stats.csv | head -1) > gpurun_out/r06_cfg5_p8_rocprof_kernel_stats.txt
rm -rf gpurun_out/r06_prof_cfg5
head -12 gpurun_out/r06_bench_default_rocprof_kernel_stats.txt; head -12 gpurun_out/r06_cfg5_p8_rocprof_kernel_stats.txt
fi
if [ $STEP = all ] || [ $STEP = pmc_loop ]; then
# 3. PMC over the headline loop's own launches (plane GEMM + plane attention)   -> r06_pgemm_loop_pmc.json, r06_attention_planes_d108_loop_pmc.json
python3 tools/pmc_collect.py gpurun_out/r06_pgemm_loop_pmc.json pgemm_kernel,attention_planes_kernel=gpurun_out/r06_attention_planes_d108_loop_pmc.json -- python3 bench.py --breakdown-only --steps 1 --warmup 1 > gpurun_out/r06_pmc_loop.log 2>&1
tail -3 gpurun_out/r06_pmc_loop.log
fi
if [ $STEP = all ] || [ $STEP = pmc_cfg ]; then
# 4. PMC over cfg3's and cfg5's plane GEMM / plane attention launches           -> r06_cfg3_*_pmc.json, r06_cfg5_*_pmc.json
python3 tools/pmc_collect.py gpurun_out/r06_cfg3_pgemm_pmc.json "pgemm16w_kernel,pgemm_kernel<9=gpurun_out/r06_cfg3_pgemm_ln_pmc.json,attention_planes_kernel=gpurun_out/r06_cfg3_attention_pmc.json" -- python3 tools/bench_cfg3.py > gpurun_out/r06_pmc_cfg3.log 2>&1
python3 tools/pmc_collect.py gpurun_out/r06_cfg5_pgemm_pmc.json "pgemm_kernel<4,attention_planes_kernel=gpurun_out/r06_cfg5_attention_pmc.json" -- python3 tools/bench_2d3d.py > gpurun_out/r06_pmc_cfg5.log 2>&1
tail -3 gpurun_out/r06_pmc_cfg3.log; tail -3 gpurun_out/r06_pmc_cfg5.log
fi
