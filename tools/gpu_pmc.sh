#!/bin/bash
# instruction-mix counters of the bench step (separate --pmc passes, no trace domains): fp64 VALU instruction counts
# (the solver's real bound) and VALU lane utilisation (divergence of the k-NN kernels).  usage: bash tools/gpu_pmc.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_$TAG && mkdir -p /tmp/pmc_$TAG
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_ACTIVE_INST_VALU\|SQ_THREAD_CYCLES_VALU\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_INSTS_LDS\|SQ_INSTS_VMEM[A-Z_]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_avail.txt
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed"
i=0
for SET in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_$TAG/p$i -- $B > /tmp/pmc_$TAG/p$i.log 2>&1 || tail -5 /tmp/pmc_$TAG/p$i.log
done
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py generic $(find /tmp/pmc_$TAG -name '*counter_collection.csv') > gpurun_out/${TAG}_pmc_valu.csv
cat gpurun_out/${TAG}_avail.txt | tr '\n' ' '; echo; head -16 gpurun_out/${TAG}_pmc_valu.csv
