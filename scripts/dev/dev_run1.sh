timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adaptors.py tests/test_gpu_fullsize.py -q -m gpu --timeout 900 -k "kerr or Kerr or config5 or golden or frame" 2>&1 | tail -2
bash scripts/ab.sh "--workload frame --rhs kerr --steps 60 --warmup 5" base prev base prev | awk '{print $1,$2,$3,$4,$5}'
