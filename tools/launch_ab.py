#!/usr/bin/env python3
"""Launch-policy experiments on ONE GPU, one tool (round 3: replaces ab_options.py, share_sweep.py, slab_scaling.py,
ab_variants.sh, ab_build_variants.py, sharded_host_timing.py).  Times device-resident frames (or one rank's cyclic share of a
frame) through rtgr_trace_rows_device_* under sets of launch options (rtgr_set_option: none of them changes a result bit)
and/or alternative builds of the library.

    python tools/launch_ab.py ab     --sets "far4=0;far4=1;fair=0,qchunk=8" [--size 4096] [--shares 1,8] [--variants ks_ref0,ks_true08]
        every set (and the default) in INTERLEAVED rounds, best round reported: box-to-box and minute-to-minute clock
        differences (±2.5 % on this pool) cancel.  ms per frame, FAR / NEAR / set-up / resolve split of the last round.
    python tools/launch_ab.py sweep  [--size 1024] [--shares 1] [--variants ks_ref0]
        one option at a time over the library's knobs (the table DESIGN §4.2 quotes)
    python tools/launch_ab.py scaling [--size 4096] [--variants ks_ref0]
        per-rank time of a cyclic N-way split, N = 1, 2, 4, 8 (a REHEARSAL of the strong-scaling curve: compute only, one GPU)
    python tools/launch_ab.py builds --flags "name1:-DA=1,-DB=2;name2:-DC=3" [--size ...]
        builds raytracegr.jl_amd/build/variants/librtgr_<name>.so with the extra flags and times each (fresh process per
        library, RTGR_LIB) against the default build with `bench.py --cpu-sample 0 --extras 0`
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWEEP = (("far4", (0, 1)), ("waves_per_cu", (4, 8, 12, 16)), ("fair", (0, 12, 13, 14)), ("qchunk", (4, 8, 16, 32, 64, 256)),
         ("waves_per_cu_near", (4, 8, 12)), ("qchunk_near", (32, 64, 128, 256)), ("near_early", (0, 16, 32, 128, 256)),
         ("order", (0,)), ("split", (0,)))


def parse_sets(text):
    sets = [dict((k, int(v)) for k, v in (kv.split("=") for kv in s.split(",") if kv)) for s in text.split(";")]
    return [{}] + [s for s in sets if s]


def device_runner(a):
    import torch
    from __graft_entry__ import load_package
    rt = load_package()
    from raytracegr_jl_amd import sharded
    import bench
    abi = rt._abi
    lib = abi.load()
    abi.check(lib, lib.rtgr_init(-1))
    npdt = {"f64": "float64", "f32": "float32"}[a.dtype]
    import numpy as np
    opt = rt.solver_defaults(getattr(np, npdt))

    def frame(variant, share):
        sc, cam = bench.build_scene(rt, variant)
        j0, st, nr = sharded.row_assignment(a.size, share, 0, "cyclic")
        out = {}

        def once():
            sharded.trace_rows_torch(sc, opt, cam, a.size, a.size, j0, st, nr, out=out, dtype=getattr(np, npdt))

        def timed(reps, **kw):
            with abi.options(lib, **kw):
                for _ in range(2):
                    once()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    once()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / reps * 1e3
                lib.rtgr_timing_enable(None, 0, 1)
                once()
                torch.cuda.synchronize()
                kms, kln = (ctypes.c_double * 4)(), (ctypes.c_uint64 * 4)()
                lib.rtgr_timing_read(None, 0, ctypes.byref(kms), ctypes.byref(kln))
                lib.rtgr_timing_enable(None, 0, 0)
            return ms, {"setup": kms[0], "far": kms[1], "near": kms[3], "resolve": kms[2]}
        return timed, a.size * nr
    return frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["ab", "sweep", "scaling", "builds"])
    ap.add_argument("--sets", default="")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--shares", default="1")
    ap.add_argument("--variants", default="ks_ref0")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--flags", default="")
    ap.add_argument("--bench-steps", type=int, default=0, help="builds: timed passes per run (default 3, 30 up to 2048²)")
    a = ap.parse_args()
    variants = a.variants.split(",")
    if a.mode == "builds":
        import importlib.util
        spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        vdir = os.path.join(ROOT, "raytracegr.jl_amd", "build", "variants")
        os.makedirs(vdir, exist_ok=True)
        libs = [("default", None)]
        for item in [x for x in a.flags.split(";") if x]:
            name, fl = item.split(":")
            out = os.path.join(vdir, f"librtgr_{name}.so")
            b.build(extra=fl.split(","), out=out, obj_dir=os.path.join(vdir, "obj_" + name), verbose=False)
            libs.append((name, out))
        for variant in variants:
            for name, path in libs:
                env = dict(os.environ)
                env.pop("RTGR_LIB", None)
                if path:
                    env["RTGR_LIB"] = path
                steps = a.bench_steps or (30 if a.size <= 2048 else 3)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(max(1, steps // 5)), "--cpu-sample", "0",
                                    "--extras", "0", "--size", str(a.size), "--variant", variant, "--dtype", a.dtype],
                                   capture_output=True, text=True, env=env)
                line = next((json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")), None)
                if line is None:
                    print(f"{variant} [{name}] FAILED: {r.stderr[-300:]}")
                    continue
                rf = line["roofline"]
                print(f"{variant} {a.size}² [{name}]: {line['ms_per_step']:.2f} ms  steps/s {line['value']:.4g}  far {rf['far_pass_ms_per_pass']:.2f} "
                      f"near {rf['near_pass_ms_per_pass']:.2f}", flush=True)
        return
    frame = device_runner(a)
    shares = [int(x) for x in a.shares.split(",")]
    if a.mode == "scaling":
        shares = [1, 2, 4, 8]
    for variant in variants:
        t1 = None
        for share in shares:
            timed, nrays = frame(variant, share)
            reps = max(3, min(10, share * 2))
            head = f"{variant} {a.dtype} {a.size}² share 1/{share} ({nrays / 1e6:.2f} M rays)"
            if a.mode == "scaling":
                ms, k = timed(reps)
                t1 = t1 or ms
                print(f"{head}: {ms:.2f} ms per rank  speed-up over N=1 {t1 / ms:.2f}x  (far {k['far']:.2f} near {k['near']:.2f} "
                      f"setup {k['setup']:.2f} resolve {k['resolve']:.2f}) — compute only, one GPU: a rehearsal", flush=True)
            elif a.mode == "sweep":
                print(head)
                print(f"{'options':34s} total   setup  FAR    NEAR   resolve")
                for kw in [{}] + [{k: v} for k, vals in SWEEP for v in vals]:
                    try:
                        ms, k = timed(reps, **kw)
                    except Exception as e:  # noqa: BLE001
                        print(f"{str(kw):34s} ERROR {e}")
                        continue
                    print(f"{str(kw):34s} {ms:6.2f}  {k['setup']:5.2f}  {k['far']:6.2f} {k['near']:5.2f}  {k['resolve']:5.2f}", flush=True)
            else:
                sets = parse_sets(a.sets)
                res, last = {}, {}
                for _ in range(a.rounds):
                    for i, kw in enumerate(sets):
                        ms, k = timed(reps, **kw)
                        res.setdefault(i, []).append(ms)
                        last[i] = k
                base = min(res[0])
                print(f"{head}: default {base:.3f} ms (far {last[0]['far']:.2f} near {last[0]['near']:.2f}) | " +
                      "  ".join(f"{sets[i]}: {min(v):.3f} ({(min(v) / base - 1) * 100:+.1f}%, far {last[i]['far']:.2f})"
                                for i, v in res.items() if i), flush=True)


if __name__ == "__main__":
    main()
