#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_user_objects.py -q -m gpu > $O/pytest_units5.log 2>&1; echo "pytest rc=$?"
tail -n 25 $O/pytest_units5.log
