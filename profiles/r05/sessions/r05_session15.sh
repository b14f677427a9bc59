#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
( time timeout -k 10 1100 python -m pytest tests -q -m gpu ) > $O/pytest_gpu_full3.log 2>&1; echo "pytest rc=$?"
tail -n 8 $O/pytest_gpu_full3.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -n 5
