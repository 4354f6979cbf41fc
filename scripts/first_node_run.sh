#!/bin/bash
# The first lease of a multi-GPU node, in one command (see scripts/first_node_run.py): every step in a fresh child
# process, one JSON line per step in gpurun_out/node_run.jsonl, failures do not stop later steps, exit code != 0 if any failed.
#   bash scripts/first_node_run.sh            # 8 GPUs
#   bash scripts/first_node_run.sh --gpus 4
#   bash scripts/first_node_run.sh --standin --gpus 2 --steps 10   # the same plan on a one-GPU box (gloo ranks / repeated device)
cd "$(dirname "$0")/.." || exit 2
mkdir -p gpurun_out
python3 scripts/first_node_run.py "$@"
rc=$?
echo "node run finished with code $rc; records: gpurun_out/node_run.jsonl"
exit $rc
