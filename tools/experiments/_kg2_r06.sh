# two key groups in the plane attention as the default? the whole GPU suite with it on, then cfg3 / cfg5 A/B
export DR_DIAGNOSTICS=1
DR_ATTN_KG2=1 timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_kg2_pytest_gpu.txt
cat gpurun_out/r06_kg2_pytest_gpu.txt
for i in 1 2; do
for v in 0 1; do
DR_ATTN_KG2=$v python tools/bench_cfg3.py 2>&1 | grep -o '"gpu_pairs_per_s": [0-9.]*' | sed "s/^/cfg3 kg2=$v /"
DR_ATTN_KG2=$v P=8 python tools/bench_2d3d.py 2>&1 | grep -o '"gpu_pairs_per_s": [0-9.]*' | head -1 | sed "s/^/cfg5 kg2=$v /"
done; done
