#!/bin/bash
# round 6, VERDICT task 1: the driver's LITERAL command (python3 bench.py --gpus 1 --steps 20 --warmup 5) five times per
# variant on one lease, the variants interleaved: calibration probes in front of the warm-up steps (round 5's protocol),
# after the timed region only (round 6's default), and --lean (no probes, no counters, nothing but the region).
# One summary line per run into gpurun_out/r06_driver_cmd_<tag>.log, the full JSON lines beside it.
tag=${1:-a}
mkdir -p gpurun_out
log=gpurun_out/r06_driver_cmd_$tag.log
full=gpurun_out/r06_driver_cmd_$tag.jsonl
: > $log; : > $full
echo "# $(date -u +%FT%TZ) $(python3 -c 'import socket; print(socket.gethostname())') $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)" >> $log
for i in 1 2 3 4 5; do
  for v in before_warmup after_only lean; do
    if [ $v = lean ]; then
      line=$(timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --lean 2>gpurun_out/r06_driver_cmd_$tag.err | tail -1)
    else
      line=$(BHGEO_PROBE_WHEN=$v timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r06_driver_cmd_$tag.err | tail -1)
    fi
    echo "$line" >> $full
    if [ -z "$line" ]; then echo "$v: no line; stderr:" >> $log; tail -15 gpurun_out/r06_driver_cmd_$tag.err >> $log; cat $log; exit 1; fi
    echo "$line" | python3 scripts/r06_line_summary.py "$v" >> $log
  done
done
cat $log
