#!/bin/bash
# One GPU-box round: parity tests, bench, rocprofv3 kernel trace of the same bench command,
# then two separate PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
# Usage (on the box, from the repo root): bash scripts/gpu_round.sh <tag> [bench args]
tag=${1:-r1}; shift
BARGS="$@"
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu_$tag.log
timeout 300 python bench.py $BARGS 2>&1 | tail -1 | tee gpurun_out/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag /tmp/pmcf_$tag /tmp/pmcw_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-seconds 0 $BARGS > $R/gpurun_out/rocprof_$tag.log 2>&1
for f in $(find /tmp/prof_$tag -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/kernel_stats_$tag.csv; done
head -4 $R/gpurun_out/kernel_stats_$tag.csv | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcf_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 $BARGS >> $R/gpurun_out/rocprof_$tag.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 $BARGS >> $R/gpurun_out/rocprof_$tag.log 2>&1
for f in $(find /tmp/pmcf_$tag -name "*counter_collection.csv"); do head -1 $f > $R/gpurun_out/pmc_fetch_$tag.csv; grep trace_ $f >> $R/gpurun_out/pmc_fetch_$tag.csv; done
for f in $(find /tmp/pmcw_$tag -name "*counter_collection.csv"); do head -1 $f > $R/gpurun_out/pmc_write_$tag.csv; grep trace_ $f >> $R/gpurun_out/pmc_write_$tag.csv; done
head -3 $R/gpurun_out/pmc_fetch_$tag.csv | cut -c1-300; head -3 $R/gpurun_out/pmc_write_$tag.csv | cut -c1-300
