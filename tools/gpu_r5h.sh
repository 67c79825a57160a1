#!/bin/bash
# round 5, call h: W = 8 fixed -> tests; C4; first run of the new C5 (covering queries) and C3 with roofline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_small.py tests/test_mapping_sequence.py tests/test_keyframes.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r5h_tests.log 2>&1
tail -5 gpurun_out/r5h_tests.log
timeout 600 python bench_c4.py --frames 400 > gpurun_out/r5h_c4.json 2> gpurun_out/r5h_c4.err
tail -c 1200 gpurun_out/r5h_c4.json; echo
( time timeout 1200 python bench_c5.py ) > gpurun_out/r5h_c5.json 2> gpurun_out/r5h_c5.err
tail -c 3000 gpurun_out/r5h_c5.json; echo; tail -5 gpurun_out/r5h_c5.err
( time timeout 900 python bench_c3.py ) > gpurun_out/r5h_c3.json 2> gpurun_out/r5h_c3.err
tail -c 2000 gpurun_out/r5h_c3.json; echo; tail -4 gpurun_out/r5h_c3.err
