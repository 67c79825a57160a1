#!/bin/bash
# run a selection of GPU tests; usage: bash tools/gpu_tests.sh <tag> <pytest args...>
TAG=${1:-x}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -40 gpurun_out/${TAG}_tests.log
