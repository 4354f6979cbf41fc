#!/bin/bash
# round 6, first lease: the driver's literal command A/B (task 1), then the event-coherence histograms of the -DBHG_DIAG
# build (task 2a) for config 3 (disk), Kerr + disk, config 4 (orbit) and a plain exit-sphere frame
tag=${1:-a}
bash scripts/gpu_r06_driver_cmd.sh $tag > /dev/null 2>&1
cat gpurun_out/r06_driver_cmd_$tag.log
bash scripts/gpu_diag.sh "disk diskkerr orbit exit" diag 2>&1 | tee gpurun_out/r06_coherence_$tag.log | grep -E "==|coherence|k |share of one|lane util"
