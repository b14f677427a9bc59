#!/bin/bash
# Short "long" lists (17-56 objects): no groups / groups of <= 8 / groups of <= 4, and one against two hand-back rounds — where the rules
# "4 spheres per group below 36 spheres" (rtgr_context.hip) and "two rounds from N objects on" (rtgr_pipeline.hpp) come from.
#     usage: bash tools/groups_small_lists.sh   -> profiles/r06/groups_small_lists.log
for g in 0 1 4; do echo "== RTGR_GROUPS=$g"; RTGR_GROUPS=$g python tools/objects_cost.py --size 2048 --counts 20,24,28,32,40,48,56 2>/dev/null; done
for r in 1 2; do echo "== RTGR_GROUPS=4 RTGR_ROUNDS=$r"; RTGR_GROUPS=4 RTGR_ROUNDS=$r python tools/objects_cost.py --size 2048 --counts 17,20,24,28,32 2>/dev/null; done
for r in 1 2; do echo "== ks_true08 RTGR_GROUPS=4 RTGR_ROUNDS=$r"; RTGR_GROUPS=4 RTGR_ROUNDS=$r python tools/objects_cost.py --size 2048 --variant ks_true08 --counts 17,20,24,28,32 2>/dev/null; done
