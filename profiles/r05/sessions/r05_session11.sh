#!/bin/bash
# what the round-end GPU tier costs on a box WITHOUT the unit cache (raytracegr.jl_amd/build/ is git-ignored): every run-time unit the
# tests and smoke() need is compiled there with hipcc
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
export RTGR_USER_CACHE=/tmp/rtgr_cold_cache
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke_cold.log 2>&1; echo "smoke rc=$?"; tail -n 5 $O/smoke_cold.log
( time timeout -k 10 1100 python -m pytest tests -q -m gpu -x ) > $O/pytest_gpu_cold_cache.log 2>&1; echo "pytest rc=$?"
tail -n 8 $O/pytest_gpu_cold_cache.log
