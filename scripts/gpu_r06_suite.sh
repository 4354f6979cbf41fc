#!/bin/bash
# round 6: the whole GPU suite with its printed census lines kept (-s), the node script with stand-ins (8 "devices" of one GPU),
# then the driver's literal bench command
tag=${1:-r06}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -s --timeout 900 --durations=12 > gpurun_out/${tag}_suite_full.log 2>&1
grep -E "every ray|config [2345]|Kerr off-axis|Kerr a/M|T2:|rounding flips|passed|failed|FAILED|Error|warning" gpurun_out/${tag}_suite_full.log | cut -c1-500 | tail -40
timeout 1500 bash scripts/first_node_run.sh --standin --gpus 8 --steps 20 --warmup 5 --quick --out gpurun_out/${tag}_node_run_standin_n8.jsonl > gpurun_out/${tag}_node_run_standin.log 2>&1
tail -45 gpurun_out/${tag}_node_run_standin.log | cut -c1-260
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_cmd.log 2>&1; tail -1 gpurun_out/${tag}_bench_driver_cmd.log > gpurun_out/${tag}_bench_driver_cmd.json
python3 scripts/r06_line_summary.py driver_cmd < gpurun_out/${tag}_bench_driver_cmd.json | cut -c1-400
