#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 300 tools/micro/exec_flip_repro.sh $O/efr > $O/exec_flip_repro.log 2>&1; echo "repro rc=$?" >> $O/exec_flip_repro.log
timeout -k 10 200 tools/micro/exec_flip_repro.sh $O/efr2 -DWAVES=2 -DEXTRA=40 > $O/exec_flip_repro_w2.log 2>&1; echo "repro rc=$?" >> $O/exec_flip_repro_w2.log
tail -n 12 $O/exec_flip_repro.log; tail -n 12 $O/exec_flip_repro_w2.log
rm -rf $O/efr $O/efr2 $O/probe
timeout -k 10 1000 python -m pytest tests/test_user_objects.py tests/test_truth.py tests/test_golden_fixtures.py tests/test_bench_gpu.py -q -m gpu -k "user_objects or shapes or test_user_objects or bench" > $O/pytest_user_objects.log 2>&1; echo "pytest rc=$?"
tail -n 40 $O/pytest_user_objects.log
