#!/usr/bin/env python3
"""What a longer object list costs: example2 at SIZE² with its three objects and with N objects (bench.py
--objects: the three + N - 3 small spheres on a spiral), device entry — ms per frame, FAR / NEAR pass, steps per ray — and, with
--live 1, the hardware's VALU count per wave-step (rocprofv3 --pmc in child processes, as bench.py's variants do).

    python tools/objects_cost.py [--size 2048] [--counts 3,16,64,256] [--live 0] [--variant ks_ref0]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--counts", default="3,16,64")
    ap.add_argument("--variant", default="ks_ref0")
    ap.add_argument("--live", type=int, default=0)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    import bench
    from __graft_entry__ import load_package
    rt = load_package()
    lib = rt._abi.load()
    rt._abi.check(lib, lib.rtgr_init(-1))
    rows = {}
    for n in [int(x) for x in a.counts.split(",")]:
        reps = 10 if n <= 16 else (4 if n <= 64 else 2)
        v = bench.time_variant(rt, a.variant, a.size, "f64", "closed", reps, 1, live_on=bool(a.live), nobj=n)
        r = v["roofline"]
        rows[n] = {"ms_per_pass": v["ms_per_pass"], "far_ms": r["far_pass_ms_per_pass"], "near_ms": r["near_pass_ms_per_pass"],
                   "setup_ms": r["other_kernels_ms_per_pass"]["setup_and_order"], "resolve_ms": r["other_kernels_ms_per_pass"]["resolve"],
                   "step_attempts_per_ray": v["step_attempts_per_ray"], "valu_per_wave_step": r.get("valu_per_wave_step"),
                   "frame_checksum": v["frame_checksum"]}
        b = rows[min(rows)]
        print(f"{a.variant} {a.size}² {n:4d} objects: {v['ms_per_pass']:8.2f} ms  ({v['ms_per_pass'] / b['ms_per_pass']:.2f} x)  far {r['far_pass_ms_per_pass']:7.2f}  "
              f"near {r['near_pass_ms_per_pass']:7.2f}  resolve {rows[n]['resolve_ms']:.2f}  steps/ray {v['step_attempts_per_ray']:.1f}"
              + (f"  VALU/wave-step {r['valu_per_wave_step']:.0f}" if r.get("valu_per_wave_step") else ""), flush=True)
    if a.json:
        json.dump({"variant": a.variant, "size": a.size, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
