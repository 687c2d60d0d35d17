# does one PMC pass over cfg3's loop finish, with and without the two key groups?  (bounded: 200 s each)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export P=8 CPU=0 DR_DIAGNOSTICS=1
for v in 0 1; do
export DR_ATTN_KG2=$v
t0=$(date +%s)
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES -d gpurun_out/pmc_probe_$v --output-format csv -- python3 tools/bench_cfg3.py > gpurun_out/pmc_probe_$v.log 2>&1
echo "KG2=$v rc=$? seconds=$(( $(date +%s) - t0 ))"
rm -rf gpurun_out/pmc_probe_$v
done
