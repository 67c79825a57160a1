#!/bin/bash
# whole GPU tier (incl. the reference cell-map fixtures and the key-frame sequence) + the key-frame exploration.  usage: bash tools/gpu_r4k.sh <tag>
TAG=${1:-r4k}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python tools/exp_keyframes.py > gpurun_out/${TAG}_keyframes.txt 2>&1; tail -12 gpurun_out/${TAG}_keyframes.txt | cut -c1-600
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -15 gpurun_out/${TAG}_tests.log
