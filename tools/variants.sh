#!/bin/bash
# bench every metric variant / dtype on one GPU
OUT=${1:-gpurun_out/variants.log}
for v in ks_ref0 ks_ref08 ks_true0 ks_true08 ks_true0998 mink; do
  for dt in f64 f32; do
    echo "### $v $dt" | tee -a $OUT
    python bench.py --size 2048 --steps 2 --warmup 1 --cpu-sample 0 --variant $v --dtype $dt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   steps/s %.4g  rays/s %.4g  ms/pass %.2f  steps/ray %.1f rejected %d' % (d['value'], d['rays_per_s'], d['ms_per_step'], d['step_attempts_per_pass']/d['rays'], d['rejected']))
" | tee -a $OUT
  done
done
