cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# 1. kernel stats of the default bench
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_prof_bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 > gpurun_out/r03_prof_bench.json 2> gpurun_out/r03_prof_bench.err
# 2. PMC over the loop's own launches (plane GEMM + plane attention)
python3 tools/pmc_collect.py gpurun_out/r03_pgemm_loop_pmc.json pgemm_kernel,attention_planes_kernel=gpurun_out/r03_attention_planes_pmc.json -- python3 bench.py --breakdown-only --steps 1 --warmup 1 > gpurun_out/r03_pmc_loop.log 2>&1
# 3. PMC of the persistent Sinkhorn kernel
python3 tools/pmc_collect.py gpurun_out/r03_sinkhorn_persist_pmc.json sk_fast_persist_kernel -- python3 tools/sk_one.py 4096 > gpurun_out/r03_pmc_sk.log 2>&1
# 4. kernel stats of B = 1
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_prof_b1 --output-format csv -- python3 tools/b1_one.py > gpurun_out/r03_prof_b1.log 2>&1
ls gpurun_out/r03_prof_bench gpurun_out/r03_prof_b1 | head
