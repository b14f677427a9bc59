#!/usr/bin/env python3
"""Experiment (round 5): can ONE small frame gain what two frames in flight gain?  A frame is traced whole on one stream, and as K
interleaved shares (rows r, r + K, …: rtgr_trace_rows_device_*) on K streams started together, so that the thin end of one share's
passes overlaps the start of the next.  Interleaved rounds, best of each (box clocks drift by ±2.5 %).  -> stdout

    python tools/split_frame_ab.py [--size 1024] [--variants ks_ref0,ks_true08] [--ways 2,3,4] [--dtype f64]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--variants", default="ks_ref0,ks_true08")
    ap.add_argument("--ways", default="2,3,4")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    rt = load_package()
    from raytracegr_jl_amd import sharded
    import bench
    lib = rt._abi.load()
    rt._abi.check(lib, lib.rtgr_init(-1))
    npdt = np.float64 if a.dtype == "f64" else np.float32
    opt = rt.solver_defaults(npdt)
    n = a.size
    for variant in a.variants.split(","):
        sc, cam = bench.build_scene(rt, variant)
        whole = {}

        def one():
            sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, dtype=npdt, out=whole)

        plans = {}
        for k in [int(x) for x in a.ways.split(",")]:
            streams = [torch.cuda.Stream() for _ in range(k)]
            outs = [{} for _ in range(k)]
            shares = [sharded.row_assignment(n, k, r, "cyclic") for r in range(k)]

            def split(streams=streams, outs=outs, shares=shares):
                for st, o, (j0, js, nr) in zip(streams, outs, shares):
                    with torch.cuda.stream(st):
                        sharded.trace_rows_torch(sc, opt, cam, n, n, j0, js, nr, dtype=npdt, out=o)
            plans[k] = (split, outs, shares)

        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fn()
                torch.cuda.synchronize()      # ONE frame at a time: latency, not a render loop
            return (time.perf_counter() - t0) / a.reps * 1e3

        best = {"whole": 1e9, **{k: 1e9 for k in plans}}
        for _ in range(a.rounds):
            best["whole"] = min(best["whole"], timed(one))
            for k, (split, _, _) in plans.items():
                best[k] = min(best[k], timed(split))
        # the shares, put back in place, are the whole frame
        one()
        torch.cuda.synchronize()
        for k, (split, outs, shares) in plans.items():
            split()
            torch.cuda.synchronize()
            full = torch.empty_like(whole["rgb"]).view(3, n, n)
            for o, (j0, js, nr) in zip(outs, shares):
                full[:, j0::js, :] = o["rgb"].view(3, nr, n)
            assert torch.equal(full.view(3, n * n), whole["rgb"]), (variant, k)
        print(f"{variant} {a.dtype} {n}²: whole frame {best['whole']:.3f} ms | " +
              "  ".join(f"{k} shares on {k} streams: {best[k]:.3f} ms ({(best[k] / best['whole'] - 1) * 100:+.1f} %)" for k in plans), flush=True)


if __name__ == "__main__":
    main()
