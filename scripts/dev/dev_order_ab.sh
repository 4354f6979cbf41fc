#!/bin/bash
# tile order A/B: measured (calibration trace) vs model vs none, per workload
for w in "--workload frame" "--workload disk" "--workload orbit --steps 60 --warmup 5" "--workload frame --rhs kerr --steps 60 --warmup 5" "--workload disk --rhs kerr --steps 100 --warmup 10"; do
  echo "== $w"
  for rep in 1 2; do for o in measured model none; do
    timeout 300 python bench.py --lean $w --order $o 2>&1 | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('$o', 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'Mrays/s %.0f' % d['value'], 'frac %.3f' % d['roofline']['frac'])
except Exception as e:
    print('$o', 'FAILED', e)
"
  done; done
done
