#!/bin/bash
# The five BASELINE.json configurations on one GPU (config 1 is the CPU-runnable 200² Minkowski case; config 3 is the
# bench default).  usage: tools/configs.sh <outfile>
OUT=${1:-gpurun_out/configs.log}
# (small frames get enough passes for the clocks to settle: 2 passes of a 3 ms frame measure the ramp, not the kernel)
run() { echo "### $1" | tee -a $OUT; shift; S=2; W=1; case "$*" in *"--size 200"*|*"--size 1024"*|*"--dtype f32"*) S=30; W=6;; esac; python bench.py --cpu-sample 0 --extras 0 --steps $S --warmup $W "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   steps/s %.4g  rays/s %.4g  ms/pass %.3f  steps/ray %.1f  rejected %d  (far %.2f near %.2f ms)' % (d['value'], d['rays_per_s'], d['ms_per_step'], d['step_attempts_per_pass']/d['rays'], d['rejected'], r['far_pass_ms_per_pass'], r['near_pass_ms_per_pass']))
" | tee -a $OUT; }
run "C1 example1 Minkowski 200x200" --variant mink --size 200
run "C2 example2 as written (KS_REF a=0) 1024x1024" --variant ks_ref0 --size 1024
run "C2' Kerr-Schild a=0.8 (textbook r) 1024x1024" --variant ks_true08 --size 1024
run "C3 example2 as written 4096x4096 (bench default)" --variant ks_ref0 --size 4096
run "C3' Kerr-Schild a=0.8 4096x4096" --variant ks_true08 --size 4096
run "C4 Kerr-Schild a=0.8 2048x2048 Float32" --variant ks_true08 --size 2048 --dtype f32
run "C4' example2 as written 2048x2048 Float32" --variant ks_ref0 --size 2048 --dtype f32
run "C5 Kerr a=0.998 + thin disk 8192x8192" --variant ks_true0998_disk --size 8192
run "C3 through rtgr_trace_pixels_f64 (Array{Pixel} in/out over PCIe)" --variant ks_ref0 --size 4096 --entry pixels
run "C3 through rtgr_trace_f64 (camera on device, RGB to host)" --variant ks_ref0 --size 4096 --entry host
run "C5 through rtgr_trace_pixels_f64 (5.9 GB of pixels each way)" --variant ks_true0998_disk --size 8192 --entry pixels
run "generic dual-number RHS 2048x2048 (reference formulation)" --variant ks_ref0 --size 2048 --rhs generic
run "user metric (textbook Kerr-Schild as run-time compiled source) 2048x2048" --variant ks_true08 --size 2048 --rhs user
run "user metric typed in Kerr-Schild form (rtgr_user_ks: f and k only) 2048x2048" --variant ks_true08 --size 2048 --rhs user_ks
