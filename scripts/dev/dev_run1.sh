timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout 900 -k "beyond_one_launch" 2>&1 | tail -5
