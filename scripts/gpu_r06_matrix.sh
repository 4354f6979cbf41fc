#!/bin/bash
# round 6, VERDICT task 1: WHAT in the default line's run makes the driver's short command read 8-13 % slower than --lean?
# The driver's literal command with one ingredient removed at a time, interleaved, 3 rounds:
#   default        clock sampler (sysfs thread) + live PMC child runs before + probes after
#   nosampler      BHGEO_NO_CLOCK_SAMPLER=1
#   nopmc          --live-pmc 0
#   neither        both off (probes still on)
#   lean           --lean
tag=${1:-b}
mkdir -p gpurun_out
log=gpurun_out/r06_matrix_$tag.log
full=gpurun_out/r06_matrix_$tag.jsonl
: > $log; : > $full
echo "# $(date -u +%FT%TZ) $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)" >> $log
run() {  # name, env assignments..., -- args
  name=$1; shift
  line=$(env "$@" timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 $EXTRA 2>gpurun_out/r06_matrix_$tag.err | tail -1)
  echo "$line" >> $full
  if [ -z "$line" ]; then echo "$name: no line; stderr:" >> $log; tail -15 gpurun_out/r06_matrix_$tag.err >> $log; fi
  echo "$line" | python3 scripts/r06_line_summary.py "$name" >> $log
}
for i in 1 2 3; do
  EXTRA="" run default X=1
  EXTRA="" run nosampler BHGEO_NO_CLOCK_SAMPLER=1
  EXTRA="--live-pmc 0" run nopmc X=1
  EXTRA="--live-pmc 0" run neither BHGEO_NO_CLOCK_SAMPLER=1
  EXTRA="--lean" run lean X=1
done
cat $log
