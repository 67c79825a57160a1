#!/bin/bash
# round 5, call c: first measurements of the small solver -- Q-pipe at B = 256 and B = 2048, C4 frames/s, kernel trace of a Q-pipe step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --q-pipe --no-cpu-baseline --no-streamed --steps 6 --warmup 2 > gpurun_out/r5c_qpipe_b256.json 2> gpurun_out/r5c_qpipe_b256.err
timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --steps 3 --warmup 1 --batch 2048 --distinct-scans 256 --q-pipe-in-flight 2 > gpurun_out/r5c_qpipe_b2048.json 2> gpurun_out/r5c_qpipe_b2048.err
timeout 600 python bench_c4.py --frames 400 > gpurun_out/r5c_c4.json 2> gpurun_out/r5c_c4.err
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_r5c
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r5c/q -- python $GRAFT_REPO_ROOT/bench.py --q-pipe --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-pipeline --batch 2048 --distinct-scans 256 > /tmp/prof_r5c/q.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_r5c/q -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/r5c_qpipe_b2048_kernels_by_grid.csv
for f in r5c_qpipe_b256 r5c_qpipe_b2048 r5c_c4; do echo "== $f"; tail -c 1500 gpurun_out/$f.json; echo; tail -3 gpurun_out/$f.err | cut -c1-300; done
head -30 gpurun_out/r5c_qpipe_b2048_kernels_by_grid.csv | cut -d, -f1-3,4,6,8,9,11-15
