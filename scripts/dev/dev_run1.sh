timeout 600 python -m pytest tests/test_gpu_adaptors.py -q -m gpu --timeout 600 -k "limited" 2>&1 | grep -E "^E |^>|Error" | head -20
