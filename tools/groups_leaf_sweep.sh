#!/bin/bash
# Spheres per group of a long list (option groups = n >= 2) by list length.   usage: bash tools/groups_leaf_sweep.sh   -> profiles/r06/groups_leaf_sweep.log
for g in 1 3 4 6 12 16; do echo "== RTGR_GROUPS=$g"; RTGR_GROUPS=$g python tools/objects_cost.py --size 2048 --counts 24,32,64,128,256,512 2>/dev/null; done
