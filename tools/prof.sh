#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes for the bench (each in its own run; see the HIP guide §7).
# usage: tools/prof.sh <tag> [bench args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-sample 0 "$@" > $OUT/bench_trace.log 2>&1
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --cpu-sample 0 "$@" > $OUT/bench_pmc_$n.log 2>&1 || echo "pmc $set failed"
done
find $OUT -name "*.csv" | head -50
