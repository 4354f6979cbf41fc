#!/bin/bash
# round 6, task 1, after the fix (sampler built before the warm-up, headline repetitions unobserved but for N clock reads,
# the thread sampler in an extra repetition): the driver's literal command, default (1 read per repetition) | no reads | --lean
tag=${1:-c}
rounds=${2:-4}
mkdir -p gpurun_out
log=gpurun_out/r06_matrix2_$tag.log
full=gpurun_out/r06_matrix2_$tag.jsonl
: > $log; : > $full
echo "# $(date -u +%FT%TZ) $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)" >> $log
run() {
  name=$1; shift
  line=$(env "$@" timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 $EXTRA 2>gpurun_out/r06_matrix2_$tag.err | tail -1)
  echo "$line" >> $full
  if [ -z "$line" ]; then echo "$name: no line; stderr:" >> $log; tail -15 gpurun_out/r06_matrix2_$tag.err >> $log; fi
  echo "$line" | python3 scripts/r06_line_summary.py "$name" >> $log
}
for i in $(seq $rounds); do
  EXTRA="" run default X=1
  EXTRA="" run reads0 BHGEO_CLOCK_READS=0
  EXTRA="--lean" run lean X=1
done
cat $log
