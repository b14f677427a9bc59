#!/bin/bash
# A/B of the experiment builds of tools/ab_build_variants.py against the default library (selected with RTGR_LIB).
summ() { python - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-34s steps/s %.4g ms %.2f far %.2f near %.2f'%(sys.argv[2], d['value'], d['ms_per_step'], r['far_pass_ms_per_pass'], r['near_pass_ms_per_pass']))
PY
}
O=gpurun_out/r02_ab; mkdir -p $O
V=raytracegr.jl_amd/build/variants
for lib in default ldsk_gen2 ldsk_gen3 gen3; do
  if [ $lib = default ]; then unset RTGR_LIB; else export RTGR_LIB=$PWD/$V/librtgr_$lib.so; fi
  python bench.py --steps 3 --warmup 1 --rhs generic --size 2048 --cpu-sample 0 --extras 0 > $O/gen_$lib.log 2>&1; summ $O/gen_$lib.log "generic ks_ref0 2048 [$lib]"
  python bench.py --steps 3 --warmup 1 --rhs generic --size 2048 --variant ks_true08 --cpu-sample 0 --extras 0 > $O/gen8_$lib.log 2>&1; summ $O/gen8_$lib.log "generic ks_true08 2048 [$lib]"
done
for lib in default ldsk_spin4 spin4; do
  if [ $lib = default ]; then unset RTGR_LIB; else export RTGR_LIB=$PWD/$V/librtgr_$lib.so; fi
  python bench.py --steps 3 --warmup 1 --variant ks_true08 --cpu-sample 0 --extras 0 > $O/spin_$lib.log 2>&1; summ $O/spin_$lib.log "closed ks_true08 4096 [$lib]"
  python bench.py --steps 3 --warmup 1 --variant ks_true08 --size 1024 --cpu-sample 0 --extras 0 > $O/spin1k_$lib.log 2>&1; summ $O/spin1k_$lib.log "closed ks_true08 1024 [$lib]"
done
