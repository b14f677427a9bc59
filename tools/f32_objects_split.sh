#!/bin/bash
# Float32 object lists by length, single FULL pass (RTGR_SPLIT=0) against FAR + NEAR (RTGR_SPLIT=1): where the automatic rule
# (RTGR_F32_SPLIT_FROM, rtgr_pipeline.hpp) comes from.   usage: bash tools/f32_objects_split.sh   -> profiles/r06/f32_objects_split_b.log
for v in ks_ref0 ks_true08; do for n in 16 24 32 48 64; do for sp in 0 1; do
  RTGR_SPLIT=$sp python bench.py --cpu-sample 0 --extras 0 --live-counters 0 --dtype f32 --size 2048 --variant $v --objects $n 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v objects $n split $sp: %8.3f ms  checksum %s' % (d['ms_per_step'], d.get('frame_checksum')))
"
done; done; done
