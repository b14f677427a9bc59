#!/bin/bash
# round 5, GPU session 2: user objects + probe + reproducer
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 300 tools/micro/exec_flip_repro.sh $O/efr > $O/exec_flip_repro.log 2>&1; echo "repro rc=$?" >> $O/exec_flip_repro.log
timeout -k 10 200 tools/micro/exec_flip_repro.sh $O/efr2 -DWAVES=2 -DEXTRA=40 > $O/exec_flip_repro_w2.log 2>&1; echo "repro rc=$?" >> $O/exec_flip_repro_w2.log
tail -12 $O/exec_flip_repro.log $O/exec_flip_repro_w2.log
timeout -k 10 900 python tools/r05_probe_experiment.py > $O/unit_probe.log 2>&1 || { echo "probe experiment failed"; tail -20 $O/unit_probe.log; }
cat $O/unit_probe.log | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_user_objects.py tests/test_truth.py -q -m gpu -x -k "user_objects or shapes or test_user_objects" > $O/pytest_user_objects.log 2>&1; echo "pytest rc=$?"
tail -30 $O/pytest_user_objects.log
