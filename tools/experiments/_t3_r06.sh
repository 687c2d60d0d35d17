timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_train_branches_gpu.py -q -s 2>&1 | grep -E "passed|failed|worst|taken apart|Error|assert " | head -30
