#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
bash tools/collect_profiles.sh r05 c
timeout -k 10 600 python -m pytest tests/test_user_objects.py -q -m gpu > $O/pytest_user_objects2.log 2>&1; echo "pytest rc=$?"
tail -n 8 $O/pytest_user_objects2.log
( time python bench.py --live-counters 0 --cpu-sample 0 ) > $O/bench_nolive.log 2>&1; echo "bench rc=$?"
grep "^{" $O/bench_nolive.log | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(d['variants']['user_sphere_ks_ref0_2048'], indent=1)); print(d['ms_per_step'])"
