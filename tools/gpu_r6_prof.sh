#!/bin/bash
# Round 6's committed profile set.  Every counter pass is its own rocprofv3 run (--pmc never with trace domains).
#   default line, ONE BATCH AT A TIME (bench.py --no-pipeline): kernel trace + stats, FETCH_SIZE, WRITE_SIZE, VALU counter sets, code objects -> <tag>_*
#   pipelined default loop: kernel trace -> timeline                                                                                       -> <tag>_pipelined_timeline.txt
#   Q-pipe at 2048 scans per batch: kernel trace, FETCH / WRITE                                                                            -> <tag>_qpipe_*
#   C3 (bench_c3.py) and C5 (bench_c5.py): kernel trace, FETCH / WRITE                                                                     -> <tag>_c3_*, <tag>_c5_*
# usage: LL_GIT_COMMIT=<hash> bash tools/gpu_r6_prof.sh <tag>   (bench*.py read the newest r*_pmc_*.csv; summaries carry "# build <id> commit <hash>")
TAG=${1:-r06a}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P; cd /tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed --no-pipeline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- $B > $P/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $B > $P/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- $B > $P/write.log 2>&1
i=0
for SET in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $SET --output-format csv -d $P/v$i -- $B > $P/v$i.log 2>&1 || tail -3 $P/v$i.log
done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $P/ptrace -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-streamed --no-q-pipe > $P/ptrace.log 2>&1
Q="python $R/bench.py --q-pipe --batch 2048 --distinct-scans 256 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-pipeline"
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P/qtrace -- $Q > $P/qtrace.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/qfetch -- $Q > $P/qfetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/qwrite -- $Q > $P/qwrite.log 2>&1
C3="python $R/bench_c3.py --cpu-scans 0 --in-flight 1"
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P/c3trace -- $C3 > $P/c3trace.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/c3fetch -- $C3 > $P/c3fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/c3write -- $C3 > $P/c3write.log 2>&1
C5="python $R/bench_c5.py --reps 2 --parity-queries 4"
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P/c5trace -- $C5 > $P/c5trace.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/c5fetch -- $C5 > $P/c5fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/c5write -- $C5 > $P/c5write.log 2>&1
cd "$R"
S="python tools/summarize_rocprof.py"
L=loam_livox_amd/libloamlivox_hip.so
f() { find $P/$1 -name "*$2.csv" | head -1; }
$S trace "$(f trace kernel_trace)" $L > gpurun_out/${TAG}_kernel_trace_by_grid.csv
$S codeobj $L > gpurun_out/${TAG}_code_objects.csv
$S pmc "$(f fetch counter_collection)" "$(f write counter_collection)" > gpurun_out/${TAG}_pmc_hbm_bytes.csv
$S generic $(find $P/v* -name '*counter_collection.csv') > gpurun_out/${TAG}_pmc_valu.csv
cp "$(f trace kernel_stats)" gpurun_out/${TAG}_kernel_stats_raw.csv
python tools/trace_overlap.py "$(f ptrace kernel_trace)" 50 > gpurun_out/${TAG}_pipelined_timeline.txt
tail -1 $P/ptrace.log | cut -c1-300 >> gpurun_out/${TAG}_pipelined_timeline.txt
$S trace "$(f qtrace kernel_trace)" $L --all > gpurun_out/${TAG}_qpipe_kernel_trace_by_grid.csv
$S pmc "$(f qfetch counter_collection)" "$(f qwrite counter_collection)" > gpurun_out/${TAG}_qpipe_pmc_hbm_bytes.csv
$S trace "$(f c3trace kernel_trace)" $L > gpurun_out/${TAG}_c3_kernel_trace_by_grid.csv
$S pmc "$(f c3fetch counter_collection)" "$(f c3write counter_collection)" > gpurun_out/${TAG}_c3_pmc_hbm_bytes.csv
$S trace "$(f c5trace kernel_trace)" $L > gpurun_out/${TAG}_c5_kernel_trace_by_grid.csv
$S pmc "$(f c5fetch counter_collection)" "$(f c5write counter_collection)" > gpurun_out/${TAG}_c5_pmc_hbm_bytes.csv
for x in trace fetch qtrace c3trace c5trace c5fetch; do echo "-- $x"; tail -1 $P/$x.log | cut -c1-200; done
head -8 gpurun_out/${TAG}_kernel_trace_by_grid.csv | cut -d, -f1-3,11-15; head -5 gpurun_out/${TAG}_pmc_hbm_bytes.csv
head -6 gpurun_out/${TAG}_c3_kernel_trace_by_grid.csv | cut -c1-160; head -4 gpurun_out/${TAG}_c3_pmc_hbm_bytes.csv
