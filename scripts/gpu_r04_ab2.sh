#!/bin/bash
# A/B of the working tree's library against build/variants/libbhgeo_prev.so on the main workloads + parity tests + VALU count
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adaptors.py tests/test_gpu_fullsize.py -q -m gpu --timeout 900 -x 2>&1 | tail -3
for w in "--workload frame" "--workload disk" "--workload orbit --steps 40 --warmup 5" "--workload frame --rhs kerr --steps 60 --warmup 5" "--workload disk --rhs kerr --steps 100 --warmup 10" "--workload frame --regime rk4 --steps 20 --warmup 3"; do
  echo "== $w"
  bash scripts/ab.sh "$w" base prev base prev
done 2>&1 | tee gpurun_out/r04_ab3.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dv && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d /tmp/dv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --ramp-seconds 0 --lean > /tmp/dv.log 2>&1
python3 - <<'PY'
import csv, glob, json
v=[]
for f in glob.glob('/tmp/dv/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='SQ_INSTS_VALU' and 'trace_' in r['Kernel_Name']: v.append(float(r['Counter_Value']))
line=[l for l in open('/tmp/dv.log') if l.startswith('{')][-1]
rs=json.loads(line)['roofline']['ray_steps_per_launch']
print('SQ_INSTS_VALU per 64 ray-steps: %.1f (%d launches)' % (sum(v)/len(v)*64/rs, len(v)))
PY
