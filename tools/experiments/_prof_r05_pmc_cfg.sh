# PMC passes over the plane GEMM / plane attention launches of cfg5 (8 pairs per call) and cfg3 (8 pairs per call)   -> profiles/r05_cfg5_*_pmc.json, r05_cfg3_*_pmc.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export P=8 CPU=0
python3 tools/pmc_collect.py gpurun_out/r05_cfg5_pgemm_pmc.json "pgemm_kernel<4,attention_planes_kernel=gpurun_out/r05_cfg5_attention_pmc.json" -- python3 tools/bench_2d3d.py > gpurun_out/r05_pmc_cfg5.log 2>&1
python3 tools/pmc_collect.py gpurun_out/r05_cfg3_pgemm_pmc.json "pgemm_kernel<9,attention_planes_kernel=gpurun_out/r05_cfg3_attention_pmc.json" -- python3 tools/bench_cfg3.py > gpurun_out/r05_pmc_cfg3.log 2>&1
tail -3 gpurun_out/r05_pmc_cfg5.log; tail -3 gpurun_out/r05_pmc_cfg3.log
