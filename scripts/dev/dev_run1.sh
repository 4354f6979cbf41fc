# a build WITHOUT the inlined prepare (every form runs the prepare pass and reads per-ray records from the workspace): golden parity
BHGEO_LIB=$PWD/build/variants/libbhgeo_noinline.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "golden or disk_five or objects" 2>&1 | tail -2
