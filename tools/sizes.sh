for sz in 1024 2048 4096; do for o in 0 1; do echo "### size $sz order $o"; RTGR_ORDER=$o python bench.py --size $sz --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   steps/s %.4g  rays/s %.4g  ms/pass %.3f  far %.3f near %.3f other %s' % (d['value'], d['rays_per_s'], d['ms_per_step'], d['roofline']['far_pass_ms_avg'], d['roofline']['near_pass_ms_avg'], d['roofline']['other_kernels_ms_avg']))
"; done; done
