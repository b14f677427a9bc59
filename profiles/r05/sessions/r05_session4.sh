#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 300 tools/micro/exec_flip_repro.sh $O/efr > $O/exec_flip_repro.log 2>&1; echo "repro rc=$?" >> $O/exec_flip_repro.log
tail -n 6 $O/exec_flip_repro.log
rm -rf $O/efr
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
tail -n 30 $O/pytest_gpu_full.log
