#!/bin/bash
# The round's profile set: rocprofv3 kernel-trace stats + PMC passes for every BASELINE configuration that has a kernel of
# its own, then tools/merge_flops.py writes profiles/<round>/flops.json (keyed by the kernel-source hash).
# usage (on the GPU box): tools/collect_profiles.sh r02
R=${1:-r02}
tools/prof.sh ${R}_ks_ref0 --steps 3 --warmup 1 > /dev/null 2>&1; echo done ks_ref0
tools/prof.sh ${R}_ks_true08 --steps 3 --warmup 1 --variant ks_true08 > /dev/null 2>&1; echo done ks_true08
tools/prof.sh ${R}_f32 --steps 3 --warmup 1 --variant ks_true08 --size 2048 --dtype f32 > /dev/null 2>&1; echo done f32
tools/prof.sh ${R}_c5 --steps 2 --warmup 1 --variant ks_true0998_disk --size 8192 > /dev/null 2>&1; echo done c5
tools/prof.sh ${R}_generic --steps 3 --warmup 1 --rhs generic --size 2048 > /dev/null 2>&1; echo done generic
tools/prof.sh ${R}_generic_true08 --steps 3 --warmup 1 --rhs generic --size 2048 --variant ks_true08 > /dev/null 2>&1; echo done generic_true08
tools/prof.sh ${R}_user_true08 --steps 3 --warmup 1 --rhs user --size 2048 --variant ks_true08 > /dev/null 2>&1; echo done user_true08
