#!/bin/bash
# instruction mix, stall and LDS counters of the registrar kernels (separate --pmc passes, no trace domains).  usage: bash tools/gpu_pmc3.sh <tag> [bench args]
TAG=${1:-x}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_$TAG && mkdir -p /tmp/pmc_$TAG
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed $@"
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_$TAG/p$i -- $B > /tmp/pmc_$TAG/p$i.log 2>&1 || tail -5 /tmp/pmc_$TAG/p$i.log
done
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py generic $(find /tmp/pmc_$TAG -name '*counter_collection.csv') > gpurun_out/${TAG}_pmc_mix.csv
grep "reg_knn\|reg_solve\|kernel," gpurun_out/${TAG}_pmc_mix.csv | cut -c1-900
