#!/bin/bash
# A/B sweep of kernel variants on one GPU (interleaved in one box; rule 24 of the HIP guide).
# usage: tools/ab_bench.sh <size> <outfile> [variant-env ...]   each variant is a quoted "K=V K=V" string
SIZE=${1:-2048}
OUT=${2:-gpurun_out/ab.log}
shift 2
run() { echo "### $*" | tee -a $OUT; env "$@" python bench.py --size $SIZE --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   steps/s %.4g  rays/s %.4g  ms/pass %.2f  frac %.3f' % (d['value'], d['rays_per_s'], d['ms_per_step'], d['roofline']['frac']))
" | tee -a $OUT; }
for rep in 1 2; do
  for v in "$@"; do run $v; done
done
