timeout 2400 python -m pytest tests/test_loop_gpu.py tests/test_planes_gpu.py tests/test_errors_gpu.py -x -q 2>&1 | tail -3
bash tools/experiments/_ab_r06_widesplit.sh
