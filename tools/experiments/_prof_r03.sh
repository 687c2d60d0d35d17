# How the round-3 files under profiles/ were produced (each block = one gpurun call; outputs land in gpurun_out/ and are copied by hand).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# 1. kernel stats of the default bench                         -> r03_bench_default_rocprof_kernel_stats{,_v2}.{txt,csv} (tools/trim_stats.py)
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_prof_bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 > gpurun_out/r03_prof_bench.json 2> gpurun_out/r03_prof_bench.err
# 2. PMC over the loop's own launches (plane GEMM + plane attention)   -> r03_pgemm_loop_pmc.json, r03_attention_planes_pmc{,_v2}.json
python3 tools/pmc_collect.py gpurun_out/r03_pgemm_loop_pmc.json pgemm_kernel,attention_planes_kernel=gpurun_out/r03_attention_planes_pmc.json -- python3 bench.py --breakdown-only --steps 1 --warmup 1 > gpurun_out/r03_pmc_loop.log 2>&1
# 3. PMC of the persistent Sinkhorn kernel                     -> r03_sinkhorn_persist_pmc_traffic.json
python3 tools/pmc_collect.py gpurun_out/r03_sinkhorn_persist_pmc.json sk_fast_persist_kernel -- python3 tools/sk_one.py 4096 > gpurun_out/r03_pmc_sk.log 2>&1
# 4. kernel stats of B = 1 (256 x 256 and a real-size pair)    -> r03_b1_rocprof_kernel_stats{,_v2}.txt, r03_b1_real_size_rocprof_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_prof_b1 --output-format csv -- python3 tools/b1_one.py > gpurun_out/r03_prof_b1.log 2>&1
B1_N=564 B1_M=629 rocprofv3 --kernel-trace --stats -d gpurun_out/r03_prof_b1_real --output-format csv -- python3 tools/b1_one.py > gpurun_out/r03_prof_b1_real.log 2>&1
# 5. benches of the other configs                              -> r03_bench_default_v*.json, r03_cfg3_bench*.json, r03_cfg5_bench*.json, r03_e2e_stage_latency*.json, r03_ragged_real_size.json
python3 bench.py > gpurun_out/r03_bench_default.json
P=8 python3 tools/bench_cfg3.py | tail -1 > gpurun_out/r03_cfg3_bench.json
CPU=0 python3 tools/bench_2d3d.py | tail -1 > gpurun_out/r03_cfg5_bench.json
python3 tools/bench_e2e.py | tail -1 > gpurun_out/r03_e2e_stage_latency.json
python3 tools/bench_ragged.py                                  # writes profiles/r03_ragged_real_size.json
# 6. micro-benchmarks                                          -> r03_procrustes_large_tiles.json (+ rocprof csv), r03_sinkhorn_large_tiles.json, r03_sinkhorn_iteration_chain.json, r03_gemm_latency_vs_staged.json
python3 tools/proc_large.py                                    # writes profiles/r03_procrustes_large_tiles.json
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_prof_proc --output-format csv -- python3 tools/proc_large.py > /dev/null 2>&1; python3 tools/trace_by_grid.py gpurun_out/r03_prof_proc proc
python3 tools/sk_large.py; python3 tools/sk_iters.py
SHAPES=mid python3 tools/gemm_small.py
# 7. experiments (scratch builds, not in the product)          -> r03_pgemm_chain_experiment.json, r03_pgemm_4wave_experiment.json, r03_pgemm_overlap_ablation.json
ROWS=65536 python3 tools/pgemm_time.py
