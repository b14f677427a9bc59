#!/usr/bin/env python3
"""Print the GPU timeline of the LAST rtgr_trace_pixels_f64 call in a rocprofv3 --kernel-trace --memory-copy-trace run of
tools/pixels_timeline.py (kernels of every stream; H2D copies summarised).

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/pixels_timeline.py
    python tools/timeline_summary.py OUT
"""
import csv
import glob
import sys

d = sys.argv[1]
kf = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
mf = glob.glob(f"{d}/*/*_memory_copy_trace.csv")[0]
ev = []
for r in csv.DictReader(open(kf)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:60], r["Stream_Id"], r["Grid_Size_X"]))
h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(mf)) if "HOST_TO_DEVICE" in r["Direction"]]
ev.sort()
small = [i for i, e in enumerate(ev) if "prepare" in e[3] and e[5] == "1048576"]   # a call begins and ends with a P-sized chunk
i0 = small[-2]
t0 = ev[i0][0]
tend = max(e[1] for e in ev[i0:])
busy = 0.0
last = t0
print(f"{'start':>8s}    {'end':>8s}  {'ms':>7s}  stream  kernel (grid)")
for e in ev[i0 - 1:]:
    a, b = (e[0] - t0) / 1e6, (e[1] - t0) / 1e6
    print(f"{a:8.2f} -> {b:8.2f} ({b - a:6.2f})  s{e[4]}  {e[3]} ({e[5]})")
    if e[4] == ev[i0][4]:
        busy += (e[1] - e[0]) / 1e6
n = [c for c in h2d if t0 - 5e6 < c[0] < tend]
print(f"# compute stream: {busy:.1f} ms of kernels in a {(tend - t0) / 1e6:.1f} ms window; {len(n)} H2D copies, "
      f"{sum(b - a for a, b in n) / 1e6:.1f} ms in total, first at {(n[0][0] - t0) / 1e6:.2f} ms")
