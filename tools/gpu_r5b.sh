#!/bin/bash
# round 5, call b: the small solver's tests + the existing tests that now route through it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_small.py tests/test_gpu_voxel.py tests/test_mapping_sequence.py tests/test_gpu_reg.py -m gpu -q 2>&1 | tail -60 ) > gpurun_out/r5b_tests.log 2>&1
tail -60 gpurun_out/r5b_tests.log
