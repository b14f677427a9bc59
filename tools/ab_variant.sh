# A/B of two builds on one variant: tools/ab_variant.sh <variant> <size>   (prev = raytracegr.jl_amd/build/librtgr_hip_prev.so)
for rep in 1 2 3; do for lib in raytracegr.jl_amd/build/librtgr_hip_prev.so raytracegr.jl_amd/librtgr_hip.so; do echo "### $lib"; RTGR_LIB=$lib python bench.py --variant ${1:-ks_true08} --size ${2:-1024} --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   steps/s %.4g  ms/pass %.3f  far %.3f near %.3f' % (d['value'], d['ms_per_step'], d['roofline']['far_pass_ms_avg'], d['roofline']['near_pass_ms_avg']))
"; done; done
