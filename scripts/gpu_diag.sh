#!/bin/bash
# per-wave stamps of -DBHG_DIAG builds: bash scripts/gpu_diag.sh "frame disk" diag prevdiag
wl=$1; shift
for v in "$@"; do for w in $wl; do
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 200 python scripts/dev/dev_diag_run.py $w gpurun_out/diag_${v}_$w.bin > /dev/null 2>&1
  echo "== $v $w"; python scripts/diag_analyze.py gpurun_out/diag_${v}_$w.bin
done; done
