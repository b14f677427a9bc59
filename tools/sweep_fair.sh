for s in 1024 1448 2048; do for lib in raytracegr.jl_amd/librtgr_hip.so raytracegr.jl_amd/build/librtgr_hip_fair11.so raytracegr.jl_amd/build/librtgr_hip_fair.so raytracegr.jl_amd/build/librtgr_hip_fair15.so; do echo "size $s $lib"; RTGR_LIB=$lib python bench.py --size $s --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms/pass %.3f  far %.3f near %.3f' % (d['ms_per_step'], d['roofline']['far_pass_ms_avg'], d['roofline']['near_pass_ms_avg']))
"; done; done
