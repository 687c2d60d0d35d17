cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export P=8 CPU=0
PMC_PASS_TIMEOUT=150 timeout 900 python3 tools/pmc_collect.py gpurun_out/r06_cfg3_pgemm_pmc.json "pgemm16w_kernel,pgemm_kernel<9=gpurun_out/r06_cfg3_pgemm_ln_pmc.json,attention_planes_kernel=gpurun_out/r06_cfg3_attention_pmc.json" -- python3 tools/bench_cfg3.py > gpurun_out/r06_pmc_cfg3.log 2>&1
echo rc=$?; tail -3 gpurun_out/r06_pmc_cfg3.log
