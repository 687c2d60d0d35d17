# A/B of the wide-wave k-split inside the cfg3 loop (the library reads DR_* knobs only with DR_DIAGNOSTICS=1)
export DR_DIAGNOSTICS=1
for i in 1 2 3; do
DR_PG_KSPLITW=0 python tools/bench_cfg3.py 2>&1 | grep -o '"gpu_pairs_per_s": [0-9.]*' | sed 's/^/off /'
DR_PG_KSPLITW=1 python tools/bench_cfg3.py 2>&1 | grep -o '"gpu_pairs_per_s": [0-9.]*' | sed 's/^/on  /'
done
