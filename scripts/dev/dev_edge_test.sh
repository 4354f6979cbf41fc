timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "keeps_every_grazing" 2>&1 | grep -E "^E |assert" | head -12
