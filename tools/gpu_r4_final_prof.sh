#!/bin/bash
# round 4's committed profile set on the final build.  The profiled command is bench.py's ONE-BATCH-AT-A-TIME loop (--no-pipeline): the
# loop kernel_ms_per_step / roofline are measured in (with batches in flight, launches of different batches share the device and
# stretch each other's intervals).  Kernel trace, HBM byte counters, VALU / fp64 instruction counters, each in its own rocprofv3 pass
# (counters never with trace domains); + the code-object register table; + a kernel trace of the Q-pipe step and the timeline
# figures of the pipelined default loop.  usage: bash tools/gpu_r4_final_prof.sh <tag>
TAG=${1:-r04a}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed --no-pipeline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- $B > /tmp/prof_$TAG/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$TAG/fetch -- $B > /tmp/prof_$TAG/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$TAG/write -- $B > /tmp/prof_$TAG/write.log 2>&1
i=0
for SET in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_BRANCH" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $SET --output-format csv -d /tmp/prof_$TAG/v$i -- $B > /tmp/prof_$TAG/v$i.log 2>&1 || tail -3 /tmp/prof_$TAG/v$i.log
done
# Q-pipe step (one batch at a time) and the pipelined default loop: kernel traces only
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG/qtrace -- python $GRAFT_REPO_ROOT/bench.py --q-pipe --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-pipeline > /tmp/prof_$TAG/qtrace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG/ptrace -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-streamed --no-q-pipe > /tmp/prof_$TAG/ptrace.log 2>&1
cd "$GRAFT_REPO_ROOT"
T=$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)
F=$(find /tmp/prof_$TAG/fetch -name '*counter_collection.csv' | head -1)
W=$(find /tmp/prof_$TAG/write -name '*counter_collection.csv' | head -1)
python tools/summarize_rocprof.py trace "$T" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_kernel_trace_by_grid.csv
python tools/summarize_rocprof.py codeobj loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_code_objects.csv
python tools/summarize_rocprof.py pmc "$F" "$W" > gpurun_out/${TAG}_pmc_hbm_bytes.csv
python tools/summarize_rocprof.py generic $(find /tmp/prof_$TAG/v* -name '*counter_collection.csv') > gpurun_out/${TAG}_pmc_valu.csv
cp $(find /tmp/prof_$TAG/trace -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_raw.csv
python tools/summarize_rocprof.py trace "$(find /tmp/prof_$TAG/qtrace -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_qpipe_kernels_by_grid.csv
python tools/trace_overlap.py "$(find /tmp/prof_$TAG/ptrace -name '*kernel_trace.csv' | head -1)" 50 > gpurun_out/${TAG}_pipelined_timeline.txt
tail -1 /tmp/prof_$TAG/ptrace.log | cut -c1-300 >> gpurun_out/${TAG}_pipelined_timeline.txt
head -12 gpurun_out/${TAG}_kernel_trace_by_grid.csv | cut -d, -f1-3,11-15; head -6 gpurun_out/${TAG}_pmc_hbm_bytes.csv; head -4 gpurun_out/${TAG}_pmc_valu.csv | cut -c1-300; head -6 gpurun_out/${TAG}_qpipe_kernels_by_grid.csv | cut -d, -f1-3,11-15; cat gpurun_out/${TAG}_pipelined_timeline.txt
