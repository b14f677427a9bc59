#!/bin/bash
# A/B sweep of kernel variants on one GPU (interleaved in one box; rule 24 of the HIP guide).
# usage: tools/ab_bench.sh <size> <outfile>
SIZE=${1:-2048}
OUT=${2:-gpurun_out/ab.log}
run() { echo "### $*" | tee -a $OUT; env "$@" python bench.py --size $SIZE --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   steps/s %.4g  rays/s %.4g  ms/pass %.2f  frac %.3f' % (d['value'], d['rays_per_s'], d['ms_per_step'], d['roofline']['frac']))
" | tee -a $OUT; }
for rep in 1 2; do
run RTGR_KERNEL=tile
run RTGR_KERNEL=persistent RTGR_THRESH=1
run RTGR_KERNEL=persistent RTGR_THRESH=4
run RTGR_KERNEL=persistent RTGR_THRESH=8
run RTGR_KERNEL=persistent RTGR_THRESH=16
run RTGR_LIB=raytracegr.jl_amd/build/librtgr_hip_wps1.so RTGR_WAVES_PER_CU=4 RTGR_THRESH=4
run RTGR_LIB=raytracegr.jl_amd/build/librtgr_hip_wps1.so RTGR_WAVES_PER_CU=4 RTGR_THRESH=8
done
