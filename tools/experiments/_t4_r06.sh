python tests/debug_stress_head_steps.py 2>&1 | grep -E "END TO END|sinkhorn backward"
python tools/train_step_time.py 2>&1 | tail -2
DR_DIAGNOSTICS=1 DR_SKB_F32=1 python tools/train_step_time.py 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_train_gpu.py -q -s 2>&1 | grep -E "passed|failed|worst|taken apart|Error|assert "
