timeout 900 python -m pytest tests/test_gpu_frame_object.py -q -m gpu --timeout 600 2>&1 | tail -3
for g in copy peer; do BHGEO_DEVICES=0,0,0,0 timeout 300 python bench.py --single-process --gpus 4 --frame-gather $g --steps 50 --warmup 5 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$g', 'value %.0f ms %.3f root_ms %.3f root_share %s' % (d['value'], d['ms_per_step'], c['root_gather_assembly_ms'], c['root_share']), c['collective'], [round(v,3) for v in c['trace_call_ms_per_device']], 'strong', round(d['strong']['value']), round(d['strong']['ms_per_step'],3))
"; done
timeout 300 python -m pytest tests/test_gpu_rccl.py -q -m gpu --timeout 600 -k single_process 2>&1 | tail -2
