timeout 900 python -m pytest tests/test_train_gpu.py -x -q -s -k "stress_head_gradient" 2>&1 | tail -15
python tools/experiments/cfg3_split_streams.py 2>&1 | tail -30
