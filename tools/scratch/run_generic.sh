#!/bin/bash
# quick A/B of the generic-path variants
summ() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-28s steps/s %.4g ms %.2f far %.2f near %.2f ref-eq TF %.1f (%.3f of peak)'%(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['far_pass_ms_per_pass'], r['near_pass_ms_per_pass'], r['reference_equivalent_tflops'], r['reference_equivalent_tflops']/78.6))
PY
}
O=gpurun_out/r02_gen; mkdir -p $O
python bench.py --steps 3 --warmup 1 --rhs generic --size 2048 --cpu-sample 0 --extras 0 > $O/gen_ref0.log 2>&1; summ $O/gen_ref0.log
python bench.py --steps 3 --warmup 1 --rhs generic --size 2048 --variant ks_true08 --cpu-sample 0 --extras 0 > $O/gen_true08.log 2>&1; summ $O/gen_true08.log
python bench.py --steps 3 --warmup 1 --rhs user --size 2048 --variant ks_true08 --cpu-sample 0 --extras 0 > $O/user_true08.log 2>&1; summ $O/user_true08.log
python bench.py --steps 3 --warmup 1 --rhs generic --size 2048 --dtype f32 --cpu-sample 0 --extras 0 > $O/gen_f32.log 2>&1; summ $O/gen_f32.log
