timeout 2400 python -m pytest tests/test_gpu_round6.py -q -m gpu -k "lookahead" 2>&1 | grep "Max abs\|Max rel\|passed\|failed\|FAILED\|Error" | tail -12
