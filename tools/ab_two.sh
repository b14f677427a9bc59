# A/B of two named builds: tools/ab_two.sh <libA> <libB> <size> [variant]
for rep in 1 2 3; do for lib in $1 $2; do echo "### $lib"; RTGR_LIB=$lib python bench.py --variant ${4:-ks_ref0} --size ${3:-4096} --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   steps/s %.4g  ms/pass %.3f  far %.3f near %.3f' % (d['value'], d['ms_per_step'], d['roofline']['far_pass_ms_avg'], d['roofline']['near_pass_ms_avg']))
"; done; done
