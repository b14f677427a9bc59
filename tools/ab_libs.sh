#!/bin/bash
# A/B of two builds of the library in fresh processes, interleaved (box-to-box clocks differ by ±2.5 % on this pool; only same-box,
# same-minute pairs compare):   tools/ab_libs.sh <other librtgr .so> [rounds] [bench.py args …]
OTHER=$1; ROUNDS=${2:-2}; shift 2
for i in $(seq $ROUNDS); do
  for which in default other; do
    if [ $which = other ]; then export RTGR_LIB=$OTHER; else unset RTGR_LIB; fi
    python bench.py --cpu-sample 0 --extras 0 --live-counters 0 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$which %-40s %8.3f ms  far %7.2f near %6.2f  checksum %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], r['far_pass_ms_per_pass'], r['near_pass_ms_per_pass'], d.get('frame_checksum')))
" "$@"
  done
done
