# How the round-5 files under profiles/ were produced (one gpurun call; outputs land in gpurun_out/ and are copied by hand).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# 1. kernel stats of the default bench                         -> r05_bench_default_rocprof_kernel_stats.{txt,csv} (tools/trim_stats.py)
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_prof_bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-other-configs > gpurun_out/r05_prof_bench.json 2> gpurun_out/r05_prof_bench.err
cp $(ls gpurun_out/r05_prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/r05_bench_default_rocprof_kernel_stats.csv
python3 tools/trim_stats.py gpurun_out/r05_bench_default_rocprof_kernel_stats.csv > gpurun_out/r05_bench_default_rocprof_kernel_stats.txt
rm -rf gpurun_out/r05_prof_bench
# 2. PMC over the loop's own launches (plane GEMM + plane attention)   -> r05_pgemm_loop_pmc.json, r05_attention_planes_pmc.json
python3 tools/pmc_collect.py gpurun_out/r05_pgemm_loop_pmc.json pgemm_kernel,attention_planes_kernel=gpurun_out/r05_attention_planes_pmc.json -- python3 bench.py --breakdown-only --steps 1 --warmup 1 > gpurun_out/r05_pmc_loop.log 2>&1
