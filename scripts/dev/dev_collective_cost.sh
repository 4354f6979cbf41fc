for m in 0 1; do
  if [ $m = 1 ]; then export BHGEO_FORCE_COLLECTIVE=1; else unset BHGEO_FORCE_COLLECTIVE; fi
  python bench.py --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['roofline']['kernel_ms'], b['config']['collective'])"
done
cd /tmp && export TMPDIR=/tmp && export BHGEO_FORCE_COLLECTIVE=1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 3 --cpu-seconds 0 > /dev/null 2>&1; cat $(find /tmp/ktc -name "*kernel_stats.csv") | cut -c1-150 | head -12
