#!/bin/bash
# rocprofv3 kernel-trace stats + PMC passes for the bench (each in its own run: --pmc is never combined with a trace).
# usage: tools/prof.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/{summary.txt, entry.json, trace/, pmc_*/}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 $R/bench.py --cpu-sample 0 --extras 0 --emit-row-checksums "$@" > $OUT/bench_plain.log 2>&1
export RTGR_NO_COMPILE=1   # (--rhs user: the plain run above has filled the cache; never start hipcc under the profiler)
export RTGR_UNIT_PROBE=0   # (… and probed the unit: under the profiler the probe's small frames would be counted as dispatches of the workload)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-sample 0 --extras 0 "$@" > $OUT/bench_trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -- python3 $R/bench.py --cpu-sample 0 --extras 0 "$@" > $OUT/bench_pmc_$i.log 2>&1 || echo "pmc $set failed"
done
python3 $R/tools/prof_summary.py $OUT --entry-out $OUT/entry.json | tee $OUT/summary.txt
# keep what is worth committing small: drop the raw per-dispatch csvs, keep summary + entry + kernel stats
find $OUT -name "*_kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*_agent_info.csv" -delete 2>/dev/null
