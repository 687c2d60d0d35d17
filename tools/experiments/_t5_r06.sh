timeout 900 python -m pytest tests/test_planes_gpu.py -q -k "k_split or wide" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_loop_gpu.py -q -k "cfg3 or k_split" 2>&1 | tail -2
for i in 1 2 3; do python tools/bench_cfg3.py 2>&1 | grep -o '"gpu_pairs_per_s": [0-9.]*'; done
