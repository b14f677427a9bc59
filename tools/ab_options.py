#!/usr/bin/env python3
"""Same-run A/B of launch options (rtgr_set_option) on one rank's cyclic share of a 4096² frame split N ways: every
option set is timed in interleaved rounds and the best round is reported, so that box-to-box and minute-to-minute clock
differences (±2.5 % on this pool) cancel.  ms per frame.

    python tools/ab_options.py "far4=0;far4=1;fair=0;fair=13" [Ns=1,2,4,8] [variants=ks_ref0,ks_true08] [rounds=3]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from raytracegr_jl_amd import sharded  # noqa: E402
import bench  # noqa: E402

abi = rt._abi
sets = [dict((k, int(v)) for k, v in (kv.split("=") for kv in s.split(",") if kv)) for s in sys.argv[1].split(";")]
sets = [{}] + [s for s in sets if s]
Ns = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
variants = (sys.argv[3] if len(sys.argv) > 3 else "ks_ref0,ks_true08").split(",")
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
opt = rt.solver_defaults()
ni = nj = 4096
for variant in variants:
    sc, cam = bench.build_scene(rt, variant)
    for N in Ns:
        j0, st, nr = sharded.row_assignment(nj, N, 0, "cyclic")
        reps = max(3, min(10, N * 2))
        out, res = {}, {}
        for _ in range(rounds):
            for k, kw in enumerate(sets):
                with abi.options(lib, **kw):
                    for _ in range(2):
                        sharded.trace_rows_torch(sc, opt, cam, ni, nj, j0, st, nr, out=out)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        sharded.trace_rows_torch(sc, opt, cam, ni, nj, j0, st, nr, out=out)
                    torch.cuda.synchronize()
                    res.setdefault(k, []).append((time.perf_counter() - t0) / reps * 1e3)
        base = min(res[0])
        print(f"{variant} N={N} ({ni * nr / 1e6:.1f} M rays): default {base:.3f} | " +
              "  ".join(f"{sets[k]}: {min(v):.3f} ({(min(v) / base - 1) * 100:+.1f}%)" for k, v in res.items() if k), flush=True)
