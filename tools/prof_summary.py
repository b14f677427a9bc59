#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel durations (kernel-trace) and PMC totals per kernel."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for f in glob.glob(d + "/trace/*/*_kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for row in csv.DictReader(open(f)):
        if float(row["Percentage"]) > 0.01:
            print(f'  {row["Name"][:90]:90s} calls {row["Calls"]:>4s}  avg {float(row["AverageNs"])/1e6:9.3f} ms  {row["Percentage"]}%')
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(d + "/pmc_*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        if "integrate_kernel" in k or "integrate_far4_kernel" in k or "prepare_kernel" in k or "resolve_kernel" in k or "trace_kernel" in k or "canvas_kernel" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            agg[k]["_vgpr"] = float(row.get("VGPR_Count") or 0)
            agg[k]["_scratch"] = float(row.get("Scratch_Size") or 0)
for k, c in agg.items():
    print("== PMC totals:", k)
    for name in sorted(c):
        print(f"  {name:28s} {c[name]:.6g}")
    if "SQ_INSTS_VALU_FMA_F64" in c:
        fl = 64 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"])
        print(f"  -> f64 flops issued (x64 lanes, before lane masking): {fl:.4g}")
