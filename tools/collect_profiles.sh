#!/bin/bash
# The round's profile set: rocprofv3 kernel-trace stats + PMC passes (each --pmc set in a run of its own, tools/prof.sh) for
# every BASELINE configuration that has a kernel of its own; then, locally, tools/merge_flops.py <round> writes
# profiles/<round>/flops.json (keyed by the kernel-source hash) and row_checksums.json.
#     usage (on the GPU box): tools/collect_profiles.sh r05 [a|b]      (two halves: one gpurun call is at most 20 minutes)
R=${1:-r05}; PART=${2:-abc}
if [[ $PART == *a* ]]; then
tools/prof.sh ${R}_ks_ref0 --steps 3 --warmup 1 > /dev/null 2>&1; echo done ks_ref0
tools/prof.sh ${R}_ks_true08 --steps 3 --warmup 1 --variant ks_true08 > /dev/null 2>&1; echo done ks_true08
tools/prof.sh ${R}_f32 --steps 5 --warmup 1 --variant ks_true08 --size 2048 --dtype f32 > /dev/null 2>&1; echo done f32 "(packed two-rays-per-lane kernel)"
RTGR_PACK=0 tools/prof.sh ${R}_f32_scalar --steps 5 --warmup 1 --variant ks_true08 --size 2048 --dtype f32 > /dev/null 2>&1; echo done f32_scalar
fi
if [[ $PART == *b* ]]; then
tools/prof.sh ${R}_c5 --steps 2 --warmup 1 --variant ks_true0998_disk --size 8192 > /dev/null 2>&1; echo done c5
tools/prof.sh ${R}_generic --steps 3 --warmup 1 --rhs generic --size 2048 > /dev/null 2>&1; echo done generic
tools/prof.sh ${R}_user_true08 --steps 3 --warmup 1 --rhs user --size 2048 --variant ks_true08 > /dev/null 2>&1; echo done user_true08
tools/prof.sh ${R}_userks_true08 --steps 3 --warmup 1 --rhs user_ks --size 2048 --variant ks_true08 > /dev/null 2>&1; echo done userks_true08
fi
if [[ $PART == *c* ]]; then
tools/prof.sh ${R}_user_sphere --steps 5 --warmup 1 --size 2048 --user-sphere > /dev/null 2>&1; echo done user_sphere "(example2 with its small sphere as a user-defined object: a unit of objects for the built-in metric)"
tools/prof.sh ${R}_objects64 --steps 3 --warmup 1 --size 2048 --objects 64 > /dev/null 2>&1; echo done objects64 "(example2 + 61 small spheres: a device table, the spheres in 8 groups, two hand-back rounds)"
tools/prof.sh ${R}_objects256 --steps 3 --warmup 1 --size 2048 --objects 256 > /dev/null 2>&1; echo done objects256 "(example2 + 253 small spheres: 32 groups, one round)"
fi
