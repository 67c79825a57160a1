#!/bin/bash
# round 5, call a: the new measured-configuration tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_measured_config.py "tests/test_gpu_c3_c5.py::test_c3_mid100_deblur_20m_map_matches_oracle" -m gpu -q -x 2>&1 | tail -30 ) > gpurun_out/r5a_tests.log 2>&1
tail -30 gpurun_out/r5a_tests.log
