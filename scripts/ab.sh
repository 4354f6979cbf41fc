#!/bin/bash
# A/B several library builds with the same bench command: scripts/ab.sh "<bench args>" name1 name2 ...
args="$1"; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then lib=""; else lib="$PWD/build/variants/libbhgeo_$v.so"; fi
  BHGEO_LIB=$lib timeout 300 python bench.py --lean $args 2>&1 | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('$v', 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'Mrays/s %.0f' % d['value'], 'Gsteps/s %.2f' % (d['ray_steps_per_s']/1e9), 'frac %.3f' % d['roofline']['frac'], d['config']['launch'])
except Exception as e:
    print('$v', 'FAILED', e)
"
done; done
