#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 600 python tools/split_frame_ab.py --size 1024 > $O/split_frame_ab.log 2>&1 || { echo failed; tail -5 $O/split_frame_ab.log; }
timeout -k 10 300 python tools/split_frame_ab.py --size 2048 --ways 2,4 --reps 5 >> $O/split_frame_ab.log 2>&1
timeout -k 10 300 python tools/split_frame_ab.py --size 2048 --ways 2,4 --dtype f32 --variants ks_true08 >> $O/split_frame_ab.log 2>&1
grep -v amdgpu.ids $O/split_frame_ab.log
