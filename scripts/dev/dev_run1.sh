timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout 900 -k "config5" -s 2>&1 | grep -E "config 5|passed|failed|^E " | head
