#!/usr/bin/env python3
"""The table of profiles/<round>/README.md from the round's entries (tools/merge_flops.py): one row per configuration directory.

    python tools/profile_table.py r06
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
ORDER = ["ks_ref0", "ks_true08", "c5", "f32", "f32_scalar", "generic", "user_true08", "userks_true08", "user_sphere", "objects64", "objects256"]
print("| config | integrate kernels: FAR (or FULL) + NEAR, average ms (kernel trace) | steps/s | ms/frame | VALU / wave-step (FMA/MUL/ADD f64) | "
      "executed flop / step attempt | executed fraction of peak | VALU busy · clock GHz | HBM B/ray |")
print("|---|---|---|---|---|---|---|---|---|")
for cfg in ORDER:
    d = os.path.join(ROOT, "profiles", rnd, cfg)
    if not os.path.isdir(d):
        continue
    e = json.load(open(os.path.join(d, "entry.json")))
    b = json.load(open(os.path.join(d, "bench_line.json")))
    kern, total_ms = [], 0.0
    for row in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        name = row.get("Name") or row.get("KernelName") or ""
        if "integrate" in name:
            avg = float(row.get("AverageNs") or row.get("Average") or 0) / 1e6
            short = name.replace("void rtgr::", "").split("(")[0]
            kern.append((avg, f"`{short}` {avg:.2f}" + (f" × {int(row['Calls']) // e['passes_profiled']}" if int(row["Calls"]) >= 2 * e["passes_profiled"] else "")))
            total_ms += float(row["TotalDurationNs"]) / 1e6 / e["passes_profiled"]
    kern.sort(reverse=True)
    f32 = b["dtype"] == "f32"
    peak = 157.3 if f32 else 78.6
    # (bench.py's roofline definition: the integrate kernels' executed flops per pass over THEIR time, from the kernel trace)
    tf = e["flop_per_step_attempt"] * e["step_attempts_per_pass"] / (total_ms * 1e-3) / 1e12
    pw = e.get("per_wave_step", {})
    mix = f'{pw.get("valu", 0):.0f}' + ("" if f32 else f' ({pw.get("fma", 0):.0f}/{pw.get("mul", 0):.0f}/{pw.get("add", 0):.0f})')
    clk = f' · {e["clock_ghz"]:.2f}' if "clock_ghz" in e else ""
    print(f'| `{cfg}` | {" + ".join(k for _, k in kern)} | {b["value"] / 1e10:.2f}·10¹⁰ | {b["ms_per_step"]:.1f} | {mix} | '
          f'{e["flop_per_step_attempt"]:.0f} | {tf:.1f} TF = **{tf / peak:.3f}** of {peak} | {e["valu_busy"]:.2f}{clk} | {e["hbm_bytes_per_ray"]:.0f} |')
