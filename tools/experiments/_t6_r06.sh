timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_models_api_gpu.py tests/test_train_branches_gpu.py -q 2>&1 | tail -3
python tools/train_step_time.py 2>&1 | tail -2
python tools/experiments/train_whole_time.py 2>&1 | tail -1 | cut -c1-400
