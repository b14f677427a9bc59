#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel durations (kernel-trace) and PMC totals per kernel, and derive
the flops.json ENTRY bench.py keys its roofline on:

    python tools/prof_summary.py gpurun_out/prof_<tag> [--entry-out entry.json]

    flop_per_step_attempt = 64 x (2 FMA + MUL + ADD wave-instructions of the integrate kernels, FAR + NEAR)
                            / (step attempts per pass x passes)          [f64 or f32 counters by the run's dtype]
    hbm_bytes_per_ray     = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / passes / rays   (all pipeline kernels; FETCH_SIZE
                            doubled as MI355X_MICROARCH.md's HBM section prescribes for wide streaming reads on gfx950)
"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
entry_out = sys.argv[sys.argv.index("--entry-out") + 1] if "--entry-out" in sys.argv else None
WANT = ("integrate_kernel", "integrate2_kernel", "integrate_far4_kernel", "prepare_kernel", "resolve_kernel", "trace_kernel", "canvas_kernel",
        "rtgr_user_integrate", "rtgr_user_prepare", "order_scatter")
stats = {}
for f in glob.glob(d + "/trace/*/*_kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats)")
    for row in csv.DictReader(open(f)):
        stats[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e6)
        if float(row["Percentage"]) > 0.01:
            print(f'  {row["Name"][:90]:90s} calls {row["Calls"]:>4s}  avg {float(row["AverageNs"])/1e6:9.3f} ms  {row["Percentage"]}%')
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(d + "/pmc_*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-70:]
        if any(w in k for w in WANT):
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
            agg[k]["_vgpr"] = float(row.get("VGPR_Count") or 0)
            agg[k]["_scratch"] = float(row.get("Scratch_Size") or 0)
            agg[k]["_lds"] = float(row.get("LDS_Block_Size") or 0)
for k, c in agg.items():
    print("== PMC totals:", k)
    for name in sorted(c):
        print(f"  {name:28s} {c[name]:.6g}")

# ---- the bench line of the un-profiled pass in the same directory ----------------------------------------------------
line = None
for f in sorted(glob.glob(d + "/bench_plain.log")) + sorted(glob.glob(d + "/bench_trace.log")):
    for l in open(f):
        if l.startswith("{"):
            line = json.loads(l)
            break
    if line:
        break
if line is None:
    sys.exit(0)
passes = line["steps"] + line["warmup"]
attempts, rays = line["step_attempts_per_pass"], line["rays"]
f32 = line["dtype"] == "f32"
sfx = "F32" if f32 else "F64"
tot = collections.Counter()
integ = [k for k in agg if "integrate" in k]
for k in integ:
    for name, v in agg[k].items():
        tot[name] += v
fma, mul, add = tot[f"SQ_INSTS_VALU_FMA_{sfx}"], tot[f"SQ_INSTS_VALU_MUL_{sfx}"], tot[f"SQ_INSTS_VALU_ADD_{sfx}"]
wave_steps = attempts * passes / 64.0
entry = {
    "kernels": integ, "size": line["config"]["size"], "passes_profiled": passes,
    "step_attempts_per_pass": attempts, "rays": rays,
    # Float32: the packed kernel's v_pk_* instructions are counted ONCE by the per-kind counters (half their flops: calibrated with
    # tools/micro/pk_counter_probe.hip, profiles/r05/pk_counter_probe.log); SQ_INSTS_VALU_FLOPS_FP32 counts them in full
    "flop_per_step_attempt": (64.0 * tot["SQ_INSTS_VALU_FLOPS_FP32"] / (attempts * passes) if (f32 and tot["SQ_INSTS_VALU_FLOPS_FP32"])
                              else 64.0 * (2 * fma + mul + add) / (attempts * passes)),
    "flop_counter": "SQ_INSTS_VALU_FLOPS_FP32" if (f32 and tot["SQ_INSTS_VALU_FLOPS_FP32"]) else f"2 FMA + MUL + ADD of SQ_INSTS_VALU_*_{sfx}",
    "flops_counter_check": ({"per_kind": 64.0 * (2 * fma + mul + add) / (attempts * passes),
                             "flops_counter": 64.0 * tot[f"SQ_INSTS_VALU_FLOPS_FP{sfx[1:]}"] / (attempts * passes)} if tot[f"SQ_INSTS_VALU_FLOPS_FP{sfx[1:]}"] else None),
    "per_wave_step": {"valu": tot["SQ_INSTS_VALU"] / wave_steps, "fma": fma / wave_steps, "mul": mul / wave_steps,
                      "add": add / wave_steps, "trans_f64": tot["SQ_INSTS_VALU_TRANS_F64"] / wave_steps,
                      "trans_f32": tot["SQ_INSTS_VALU_TRANS_F32"] / wave_steps, "cvt": tot["SQ_INSTS_VALU_CVT"] / wave_steps,
                      "f32_fma_mul_add": (tot["SQ_INSTS_VALU_FMA_F32"] + tot["SQ_INSTS_VALU_MUL_F32"] + tot["SQ_INSTS_VALU_ADD_F32"]) / wave_steps if not f32 else None,
                      "int": (tot["SQ_INSTS_VALU_INT32"] + tot["SQ_INSTS_VALU_INT64"]) / wave_steps,
                      "salu": tot["SQ_INSTS_SALU"] / wave_steps},
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    "valu_busy": 4.0 * tot["SQ_ACTIVE_INST_VALU"] / (1024.0 * tot["GRBM_GUI_ACTIVE"] / 8.0) if tot["GRBM_GUI_ACTIVE"] else None,
    "lane_utilisation": tot["SQ_THREAD_CYCLES_VALU"] / (tot["SQ_ACTIVE_INST_VALU"] * 64.0) if tot["SQ_ACTIVE_INST_VALU"] else None,
}
allk = collections.Counter()
for k in agg:
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        allk[name] += agg[k][name]
if allk["FETCH_SIZE"] or allk["WRITE_SIZE"]:
    entry["hbm_bytes_per_ray"] = (2 * allk["FETCH_SIZE"] + allk["WRITE_SIZE"]) * 1024.0 / passes / rays
    entry["hbm_note"] = "2 x FETCH_SIZE + WRITE_SIZE over every pipeline kernel (KB x 1024), per pass, per ray"
# main integrate kernel: duration (kernel trace), clock
main_k = max(integ, key=lambda k: agg[k]["SQ_INSTS_VALU"]) if integ else None
for name, (ncalls, avg_ms) in stats.items():
    if main_k and main_k.split("<")[0].split("::")[-1] in name and ("Li1EEE" in name or "far4" in name or "Li0EEE" in name or "user_integrate_far" in name or "full" in name):
        entry.setdefault("main_kernel_avg_ms", {})[name[:80]] = avg_ms
if main_k and agg[main_k]["GRBM_GUI_ACTIVE"]:
    for name, (ncalls, avg_ms) in stats.items():
        if main_k.split("(")[0].split("::")[-1].split("<")[0] in name and abs(ncalls - passes) == 0:
            ghz = agg[main_k]["GRBM_GUI_ACTIVE"] / 8.0 / passes / (avg_ms * 1e-3) / 1e9
            if 1.0 < ghz < 2.6:
                entry["clock_ghz"] = ghz
print("== flops.json entry")
print(json.dumps(entry, indent=1))
if entry_out:
    json.dump(entry, open(entry_out, "w"), indent=1)
