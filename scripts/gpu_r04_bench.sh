#!/bin/bash
# round 4: bench.py's GPU tests + frame-object tests, then the default line and the 2-rank gloo line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_frame_object.py -q -m gpu --timeout 900 > gpurun_out/r04_bench_pytest.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_bench_pytest.log | tail -20
timeout 900 python bench.py > gpurun_out/r04_bench_default.log 2>&1; tail -1 gpurun_out/r04_bench_default.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'frac',d['roofline']['frac'],'valu/64',d['roofline']['valu_insts_per_64_ray_steps'])
for n,s in d['strong_predicted']['shards'].items(): print(n,{k:(round(v,4) if isinstance(v,float) else v) for k,v in s.items()})
"
BHGEO_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 5 --cpu-seconds 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('2 ranks gloo: value',d['value'],'root_share',d['config']['root_share'],'strong',d['strong'])
"
