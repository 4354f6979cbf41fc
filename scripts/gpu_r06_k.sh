#!/bin/bash
# round 6 lease k: the node script with stand-ins on the final tree + the driver's command
mkdir -p gpurun_out
timeout 1500 bash scripts/first_node_run.sh --standin --gpus 8 --steps 20 --warmup 5 --quick --out gpurun_out/r06_node_run_standin_n8.jsonl > gpurun_out/r06_node_run_standin.log 2>&1
tail -26 gpurun_out/r06_node_run_standin.log | cut -c1-200
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_bench_driver_cmd_final.json
python3 scripts/r06_line_summary.py driver_cmd < gpurun_out/r06_bench_driver_cmd_final.json | cut -c1-400
