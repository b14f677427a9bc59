#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_user_objects.py tests/test_unit_probe.py -q -m gpu > $O/pytest_units4.log 2>&1; echo "pytest rc=$?"
tail -n 30 $O/pytest_units4.log
python examples/render.py 5 256 > $O/render5.log 2>&1; echo "render rc=$?"; tail -n 2 $O/render5.log; cp scenes/sphere5.png $O/ 2>/dev/null
