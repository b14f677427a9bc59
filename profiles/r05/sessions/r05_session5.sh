#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU --output-format csv -d $O/pk1 -- $R/tools/micro/pk_counter_probe > $O/pk_probe1.log 2>&1 || echo "pk pass 1 failed"
rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP32 --output-format csv -d $O/pk2 -- $R/tools/micro/pk_counter_probe > $O/pk_probe2.log 2>&1 || echo "pk pass 2 failed"
python3 - <<'PY' > $O/pk_counter_probe.log 2>&1
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05")
t = collections.defaultdict(dict)
for d in ("pk1", "pk2"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            t[row["Kernel_Name"]][row["Counter_Name"]] = t[row["Kernel_Name"]].get(row["Counter_Name"], 0) + float(row["Counter_Value"])
print("one wave per kernel, 4096 instructions of the named kind (gfx950, rocprofv3 --pmc; two passes)")
for k in sorted(t):
    print(f"{k:16s} " + "  ".join(f"{c.replace('SQ_INSTS_VALU_', '').replace('SQ_INSTS_', '')}={int(v)}" for c, v in sorted(t[k].items())))
PY
cat $O/pk_counter_probe.log
rm -rf $O/pk1 $O/pk2
cd $R
timeout -k 10 900 python -m pytest tests/test_user_metric.py tests/test_user_objects.py tests/test_unit_probe.py -q -m gpu > $O/pytest_units.log 2>&1; echo "pytest rc=$?"
tail -n 25 $O/pytest_units.log
timeout -k 10 300 python bench.py --size 1024 --steps 10 --warmup 2 --cpu-sample 1024 --extras 0 --live-counters 0 > $O/cpu_baseline_1024.log 2>&1; echo "bench rc=$?"
tail -c 1500 $O/cpu_baseline_1024.log
